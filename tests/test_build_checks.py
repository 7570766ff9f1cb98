"""CPU: properties of the generated code that the kernels rely on and that no run-time test can see directly.

scan_duo.hip keeps the helper waves' row buffers in accumulator registers that it addresses by NUMBER inside asm statements
(a0 .. a127); the compiler only knows them as clobbered.  That is sound as long as nothing the compiler emits in the helper
waves' code writes an accumulator register.  This test compiles the file to ISA (hipcc cross-compiles without a GPU) and
checks it, together with the register budget (two waves per SIMD: 256 registers per lane) and the absence of spills."""
import os
import re
import subprocess

import pytest

from conftest import PKG

SRC = os.path.join(PKG, "csrc", "scan_duo.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "scan_duo.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-inline-asm", "-S",
                    "--cuda-device-only", "-o", out, SRC], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return open(out).read()


def kernels(isa):
    """name -> (body lines, metadata dict)"""
    meta = {}
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)"
                         r".*?\.vgpr_spill_count:\s+(\d+)", isa, flags=re.S):
        meta[m.group(2)] = dict(agpr=int(m.group(1)), scratch=int(m.group(3)), vgpr=int(m.group(4)), spill=int(m.group(5)))
    bodies = {}
    for name in meta:
        a = isa.index("\n%s:" % name)
        b = isa.index("s_endpgm", a)
        b = isa.index(".end_amdhsa_kernel", b) if ".end_amdhsa_kernel" in isa[b:] else len(isa)
        bodies[name] = isa[a:b].split("\n")
    return bodies, meta


def test_duo_register_budget_and_no_spills(isa):
    bodies, meta = kernels(isa)
    duo = {k: v for k, v in meta.items() if "k_scan_duo" in k}
    assert len(duo) == 4  # C in {32, 64} x K in {2, 3}
    for name, m in duo.items():
        assert m["spill"] == 0 and m["scratch"] == 0, (name, m)
        assert m["agpr"] == 128, (name, m)
        if "ILi64E" in name:  # 512 threads: two waves per SIMD
            assert m["vgpr"] <= 256, (name, m)
        assert not any("scratch_" in ln for ln in bodies[name]), name


def test_compiler_leaves_the_row_buffers_alone(isa):
    bodies, _ = kernels(isa)
    for name, lines in bodies.items():
        if "k_scan_duo" not in name:
            continue
        in_asm = False
        row_asm, compiler_writes = [], []
        for i, ln in enumerate(lines):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm:
                # the row-buffer statements name accumulator registers directly
                if re.search(r"\ba\[(0x[0-9a-f]+|\d+):", t) and re.match(r"(ds_read_b128|ds_write_b128|global_load_dwordx4|global_store_dwordx4)", t):
                    row_asm.append(i)
                continue
            # compiler-emitted writes of accumulator registers
            if re.match(r"v_accvgpr_(write|mov)_b32", t) or re.match(r"v_mfma\S*\s+a\[", t) or re.match(r"(ds_read|global_load|buffer_load|scratch_load)\S*\s+a\[", t):
                compiler_writes.append(i)
        assert row_asm, name
        lo, hi = min(row_asm), max(row_asm)
        inside = [i for i in compiler_writes if lo <= i <= hi]
        assert not inside, "%s: the compiler writes accumulator registers between the row-buffer statements (lines %s)" % (name, inside[:8])


def test_wide_kernels_fit_their_launches(tmp_path):
    """wide.hip: the team scan must keep its register image of the folded taps without spilling (a resident team member that
    spills would pay a scratch round trip per diagonal), and a team member of 8 waves must fit two waves per SIMD; the fold
    and the weight gradient do not spill either."""
    out = str(tmp_path / "wide.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-inline-asm", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(PKG, "csrc", "wide.hip")], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE)
    _, meta = kernels(open(out).read())
    team = {k: v for k, v in meta.items() if "k_scan_team" in k}
    assert len(team) == 12  # 3 .. 8 waves x K in {2, 3}
    for name, m in meta.items():
        assert m["spill"] == 0 and m["scratch"] == 0, (name, m)
    for name, m in team.items():
        assert m["vgpr"] <= 256, (name, m)
