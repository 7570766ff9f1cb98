"""CPU: properties of the generated code that the kernels rely on and that no run-time test can see directly.

scan_duo.hip issues its LDS reads as inline asm and counts the waits by hand (hipcc answers a block of outstanding ds_reads with
lgkmcnt(0)).  The compiler believes such a read's destination is written AT the statement: if nothing reads the value later --
the last step's fragments for taps that feed a diagonal beyond the image, in a peeled copy of the step -- it hands the registers
out again at once, and the data that lands later goes on top of whatever lives there by then (seen twice in round 3: a wrong
last pixel for odd row counts, garbage z in the redo sweeps).  The source therefore names every destination in a statement
behind the wait that covers it; this test walks the ISA (hipcc cross-compiles without a GPU) with the hardware's rule -- LDS
operations of a wave complete in order, lgkmcnt(N) leaves the youngest N in flight -- and fails on any instruction that
touches a register whose read is still in flight.  Together with the register budget (two waves per SIMD: 256 registers per
lane) and the absence of spills."""
import os
import re
import subprocess

import pytest

from conftest import PKG

SRC = os.path.join(PKG, "csrc", "scan_duo.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (the flags of inverse-flow_amd/build.py for this file)
DUO_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-inline-asm", "-fno-slp-vectorize"]


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "scan_duo.s")
    subprocess.run([HIPCC] + DUO_FLAGS + ["-S", "--cuda-device-only", "-o", out, SRC], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE)
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
    assert len(duo) == 8  # C in {32, 64} x K in {2, 3} x storage in {f32, bf16}
    for name, m in duo.items():
        # (no vector register is spilled and no instruction touches scratch memory; at the register limit the backend may
        # still reserve a few dwords of frame for its scavenger -- 36 bytes in the 64-channel kernels -- which nothing uses)
        assert m["spill"] == 0 and m["scratch"] <= 64, (name, m)
        if "ILi64E" in name:  # 512 threads: two waves per SIMD
            assert m["vgpr"] <= 256, (name, m)
        assert not any(re.match(r"\s*(scratch_|buffer_(load|store))", ln) for ln in bodies[name]), name


def _regs(tok):
    """v5 -> {('v', 5)}; a[4:7] -> {('a', 4), ..., ('a', 7)}"""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)} if m else set()


def async_read_violations(lines):
    """Walk a kernel's ISA in listing order.  pending: the LDS operations in flight, oldest first, each with the registers it
    will write (a store: none).  An unconditional branch ends a straight-line stretch: what follows it in the listing is
    reached from elsewhere, with whatever that path left in flight (the walk restarts empty there: a heuristic, the listing is
    not a control-flow graph)."""
    pending, viol = [], []
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        toks = re.findall(r"[va]\[\d+:\d+\]|\b[va]\d+\b", code)
        used = set().union(*[_regs(x) for x in toks]) if toks else set()
        busy = set().union(*[p for p in pending]) if pending else set()
        if op.startswith("ds_"):
            dest = _regs(toks[0]) if (op.startswith("ds_read") and toks) else set()
            if (used - dest) & busy:
                viol.append((code, sorted((used - dest) & busy)))
            # (a read into the registers of a read still in flight is in order: the later data wins)
            pending.append(dest)
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            elif "vmcnt" not in code and "expcnt" not in code:
                pending = []
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = []
            continue
        if used & busy:
            viol.append((code, sorted(used & busy)))
    return viol


def test_no_register_is_touched_while_its_lds_read_is_in_flight(isa):
    bodies, _ = kernels(isa)
    duo = {k: v for k, v in bodies.items() if "k_scan_duo" in k}
    assert len(duo) == 8
    for name, lines in duo.items():
        n_reads = sum(1 for ln in lines if ln.strip().startswith("ds_read"))
        assert n_reads > 50, name  # (the walk looked at the right thing)
        viol = async_read_violations(lines)
        assert not viol, (name, viol[:6])


def test_the_walk_finds_a_planted_hazard():
    """the checker itself: a destination reused before the covering wait is reported, one reused after it is not"""
    bad = ["ds_read_b128 v[4:7], v20", "v_add_u32_e32 v5, 1, v9", "s_waitcnt lgkmcnt(0)"]
    good = ["ds_read_b128 v[4:7], v20", "ds_read_b128 v[8:11], v20 offset:16", "s_waitcnt lgkmcnt(1)", "v_add_u32_e32 v5, 1, v5",
            "s_waitcnt lgkmcnt(0)", "v_add_u32_e32 v8, 1, v9"]
    assert async_read_violations(bad) and not async_read_violations(good)


def asm_memory_hazards(lines):
    """Hazards nobody inserts wait states for around an inline-asm vector-memory instruction on gfx950 (each cost a
    GPU fault or a wrong store in round 2).  Walks the listing and reports, for instructions INSIDE asm statements only (the
    compiler pads its own):
      A  a scalar register written by a vector instruction (v_readlane / v_readfirstlane: a spill reload) or m0 written by
         s_mov, used by a global_/buffer_ instruction less than 5 wait states later;
      B  a store of more than 8 bytes whose data registers are written less than 2 wait states later."""
    viol = []
    in_asm = False
    recent = []  # (states ago, scalar registers written by VALU / m0)
    last_store = None  # (data regs, states since)
    def sregs(tok):
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return {int(m.group(1))}
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
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
        code = t.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        states = int(code.split()[1]) + 1 if op == "s_nop" else 1
        stoks = re.findall(r"s\[\d+:\d+\]|\bs\d+\b", code)
        vtoks = re.findall(r"[va]\[\d+:\d+\]|\b[va]\d+\b", code)
        if in_asm and (op.startswith("global_") or op.startswith("buffer_")):
            used = set().union(*[sregs(x) for x in stoks]) if stoks else set()
            for age, regs in recent:
                if age < 5 and ((used & regs) or ("m0" in regs and "lds" in op)):
                    viol.append(("A", code, age))
        if last_store is not None:
            regs, age = last_store
            written = _regs(vtoks[0]) if (vtoks and not op.startswith("global_store") and not op.startswith("ds_write")
                                          and not op.startswith("s_")) else set()
            if age < 2 and (written & regs):
                viol.append(("B", code, age))
            last_store = (regs, age + states) if age + states < 2 else None
        if in_asm and re.match(r"global_store_dwordx[34]", op):
            last_store = (_regs(vtoks[1]) if len(vtoks) > 1 else set(), 0)
        recent = [(age + states, regs) for age, regs in recent if age + states < 5]
        if op in ("v_readlane_b32", "v_readfirstlane_b32") and stoks:
            recent.append((0, sregs(stoks[0])))
        if op == "s_mov_b32" and code.split()[1].rstrip(",") == "m0":
            recent.append((0, {"m0"}))
    return viol


def test_asm_memory_instructions_keep_their_wait_states(isa, tmp_path):
    """scan_duo.hip and wide.hip: the vector-memory instructions issued from inline asm (LDS-DMA of the rows and of the mailbox
    lines, whole-line stores, the fold's hand-issued loads) keep the wait states of hazards A and B; the walk itself is
    checked on planted cases."""
    assert asm_memory_hazards([";;#ASMSTART", "s_mov_b32 m0, s3", "global_load_lds_dwordx4 v1, s[4:5]", ";;#ASMEND"])
    assert not asm_memory_hazards([";;#ASMSTART", "s_mov_b32 m0, s3", "s_nop 4", "global_load_lds_dwordx4 v1, s[4:5]", ";;#ASMEND"])
    assert asm_memory_hazards(["v_readlane_b32 s4, v9, 3", ";;#ASMSTART", "s_nop 1", "global_store_dwordx4 v1, v[4:7], s[4:5]", ";;#ASMEND"])
    assert asm_memory_hazards([";;#ASMSTART", "s_nop 4", "global_store_dwordx4 v1, v[4:7], s[4:5]", ";;#ASMEND", "v_mov_b32_e32 v5, 0"])
    assert not asm_memory_hazards([";;#ASMSTART", "s_nop 4", "global_store_dwordx4 v1, v[4:7], s[4:5]", "s_nop 1", ";;#ASMEND",
                                   "v_mov_b32_e32 v5, 0"])
    bodies, _ = kernels(isa)
    out = str(tmp_path / "wide.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-inline-asm", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(PKG, "csrc", "wide.hip")], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE)
    wbodies, _ = kernels(open(out).read())
    n_asm_mem = 0
    for name, lines in list(bodies.items()) + list(wbodies.items()):
        n_asm_mem += sum(1 for ln in lines if re.match(r"\s*(global_load_lds|global_store_dwordx4|global_load_dwordx4)", ln))
        viol = asm_memory_hazards(lines)
        assert not viol, (name, viol[:6])
    assert n_asm_mem > 20


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


def test_conditioner_kernels_do_not_spill(tmp_path):
    """conditioner.hip: the first forms of these kernels spilled up to 2.8 KB a lane (fully unrolled blocks of uniform-address
    LDS reads are hoisted to the top whatever the fences say: DESIGN 4.4b); every kernel of the file must compile without a
    spilled register or a byte of scratch, within the 256 registers two waves per SIMD leave a lane, and no load of an
    activation may sit in a divergent branch (a branch per load serialises the round trips) -- counted here as: no kernel has
    more exec-mask branches than a fixed small number."""
    out = str(tmp_path / "conditioner.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-inline-asm", "-S",
                    "--cuda-device-only", "-o", out, os.path.join(PKG, "csrc", "conditioner.hip")], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.PIPE)
    bodies, meta = kernels(open(out).read())
    cond = {k: v for k, v in meta.items() if "k_cond_" in k}
    # prep, prep_many, wreduce, wgrad x 2 operand types, and per channel count (7): fwd1m, fwd2, bwd3g, and x 2 operand types
    # bwd1, bwd2m, bwd3u
    assert len(cond) == 3 + 2 + 7 * (3 + 2 * 3), sorted(cond)
    for name, m in cond.items():
        assert m["spill"] == 0 and m["scratch"] == 0 and m["vgpr"] <= 256, (name, m)
        assert not any(re.match(r"\s*(scratch_|buffer_(load|store))", ln) for ln in bodies[name]), name
        if "k_cond_bwd1" in name or "k_cond_fwd2" in name:  # the two kernels with 9 x NI neighbourhood loads per lane
            n_br = sum(1 for ln in bodies[name] if re.match(r"\s*s_cbranch_execz", ln))
            assert n_br <= 30, (name, n_br)  # (the staging copies have a few; a branch per load would be 9 per channel)


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
