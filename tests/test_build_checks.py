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


def _vregs(tok):
    """v5 -> {5}; v[4:7] -> {4, 5, 6, 7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def test_fold_never_touches_a_register_whose_load_is_in_flight(tmp_path):
    """wide.hip, k_wide_fold: the column blocks of L arrive through loads the compiler does not see (asm) and are released by
    hand-counted waits.  Sound only if nothing between a load and the wait that covers it reads or writes its destination
    registers -- no copy made early, no reuse for something else.  Walks the kernel's ISA in program order with the hardware's
    rule (loads complete in order; vmcnt(N) leaves the youngest N in flight)."""
    out = str(tmp_path / "wide.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-inline-asm", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(PKG, "csrc", "wide.hip")], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE)
    bodies, _ = kernels(open(out).read())
    (name, lines), = [(k, v) for k, v in bodies.items() if "k_wide_fold" in k]
    in_asm, pending, n_asm_loads, n_waits = False, [], 0, 0  # pending: destination register sets, oldest first
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0]
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", code)
        used = set().union(*[_vregs(x) for x in toks]) if toks else set()
        if in_asm and code.startswith("global_load_dwordx4"):
            dest = _vregs(toks[0])
            assert not (set().union(*pending) & dest if pending else set()), (name, "load into registers of a load in flight", t)
            pending.append(dest)
            n_asm_loads += 1
            continue
        m = re.search(r"vmcnt\((\d+)\)", code)
        if code.startswith("s_waitcnt") and m:
            n = int(m.group(1))
            if in_asm:
                n_waits += 1
            # (compiler-tracked loads in flight only make the hardware wait longer than this model assumes)
            pending = pending[len(pending) - n:] if n < len(pending) else pending
            if n == 0:
                pending = []
            if in_asm:
                continue  # the wait's own register operands are the point of it
        busy = set().union(*pending) if pending else set()
        assert not (used & busy), (name, "touches registers of a load in flight", t, sorted(used & busy))
    assert n_asm_loads == 36 and n_waits == 15 + 4, (n_asm_loads, n_waits)  # 4,4,4,3,3,3,3,2,2,2,2,1,1,1,1 loads; a wait per step + 4 at the end
