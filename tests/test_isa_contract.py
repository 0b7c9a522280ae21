"""What DESIGN.md says about the BUILT device code, asserted on the assembly hipcc emits for gfx950 (CPU-only: hipcc cross-compiles,
about 10 s).  The quotes of DESIGN §6 (the occupancy mask's same-launch hand-off), the register / scratch budgets of the hot
kernels, and the source shape that keeps a known ROCm 7.2 miscompile away are contracts of the build, not of the source: a new
compiler or an innocent edit must turn this file red, not a frame wrong."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracing_octrees_amd", "csrc")


def _hipcc():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc = _hipcc()
    if not hipcc:
        pytest.skip("no hipcc in this environment")
    from ray_tracing_octrees_amd import _build
    out = tmp_path_factory.mktemp("isa") / "rto.s"
    flags = [f for f in _build.HIP_FLAGS if f not in ("-fPIC", "-shared")]           # the product's own flags
    subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", os.path.join(CSRC, "rto_api.hip"), "-o", str(out)],
                   check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def kernel_meta(asm_text):
    md = asm_text[asm_text.index(".amdgpu_metadata"):]
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size: (\d+)\n.*?\.sgpr_spill_count: (\d+)\n.*?"
                         r"\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)", md, re.S):
        out[m.group(1)] = dict(scratch=int(m.group(2)), sgpr_spill=int(m.group(3)), vgpr=int(m.group(4)), vgpr_spill=int(m.group(5)))
    return out


def find(meta, fragment):
    hits = [k for k in meta if fragment in k]
    assert hits, f"no kernel named *{fragment}* in the build"
    return hits


def body(asm_text, fragment):
    """Instructions of the one kernel whose mangled name contains `fragment` (comments, labels and directives dropped)."""
    lines = asm_text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3rto\w*" + re.escape(fragment) + r"\w*:", l)]
    assert len(starts) == 1, (fragment, len(starts))
    end = next(i for i in range(starts[0], len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins = []
    for l in lines[starts[0] + 1:end]:
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith(".") and not re.match(r"^\.?\w+:$", t):
            ins.append(t)
    return ins


def test_hot_kernels_keep_their_register_and_scratch_budgets(asm):
    meta = kernel_meta(asm)
    assert len(meta) > 60
    for exact in (0, 1):                                           # the two builds of the default kernel: general / exact-grid child tests
        lean0 = meta[find(meta, f"12k_trace_leanILi0ELi{exact}EE")[0]]
        assert lean0["scratch"] == 0 and lean0["vgpr_spill"] == 0
        assert lean0["vgpr"] <= 80                                 # 6 waves per SIMD (512 / 6 = 85, allocated by 8)
    for frag in ("k_trace_lean_batch", "22k_trace_lean_triangles", "28k_trace_lean_triangles_batch", "12k_probe_skip",
                 "24k_octree_ray_skip_packed", "13k_skip_render", "20k_closest_near_first", "11k_cull_desc", "13k_order_build",
                 "23k_trace_lean_persistent"):
        for k in find(meta, frag):
            assert meta[k]["vgpr_spill"] == 0, (k, meta[k])
            if meta[k]["scratch"]:
                # a private segment the kernel never touches (slots reserved for scalar registers that were then kept in vector-register
                # lanes): what matters is that no instruction goes to scratch
                ins = body(asm, k[len("_ZN3rto"):])
                assert not any(t.startswith(("scratch_", "buffer_load", "buffer_store")) for t in ins), (k, meta[k])
            else:
                assert meta[k]["scratch"] == 0
    for k in find(meta, "22k_trace_lean_triangles"):
        assert meta[k]["vgpr"] <= 80, (k, meta[k])                  # round 5: 6 waves per SIMD (__launch_bounds__(256, 6): 512 / 6 = 85, allocated by 8)
    for k in find(meta, "28k_trace_lean_triangles_batch"):
        assert meta[k]["vgpr"] <= 80, (k, meta[k])
    for k in find(meta, "23k_trace_lean_persistent"):
        assert meta[k]["vgpr"] <= 96, (k, meta[k])


def test_no_mfma_on_this_path(asm):
    assert "v_mfma" not in asm                                     # branchy slab arithmetic, no contraction


def _handoff(ins):
    """(stamp section, everything behind the hand-off barrier, everything in front of the stamp section): the hand-off barrier is the
    last s_barrier in front of the ticket (global_atomic_inc); the stamps are the stores between it and the barrier before it."""
    tick = [i for i, t in enumerate(ins) if t.startswith("global_atomic_inc")]
    assert len(tick) == 1, "one ticket per kernel"
    bar = max(i for i, t in enumerate(ins[:tick[0]]) if t.startswith("s_barrier"))
    prev = max(i for i, t in enumerate(ins[:bar]) if t.startswith("s_barrier"))
    return ins[prev + 1:bar], ins[bar + 1:], ins[:prev]


@pytest.mark.parametrize("kernel", ["12k_trace_leanILi0ELi1EE", "12k_trace_leanILi0ELi0EE", "22k_trace_lean_trianglesILi0ELb0EE", "13k_skip_render"])
def test_mask_handoff_has_the_gfx950_form_design_quotes(asm, kernel):
    """every stamp an sc1 store; s_waitcnt vmcnt(0) in front of the barrier; behind it lane 0: write-back, wait, ticket
    (returning atomic), wait; the last ticket: write-back, wait, sc1 flag store.  Consumer: sc1 poll, wait, sc1 load of the word."""
    ins = body(asm, kernel)
    before, behind, head = _handoff(ins)
    stores = [t for t in before if t.startswith("global_store")]
    assert stores and all(t.endswith("sc1") for t in stores), [t for t in stores if not t.endswith("sc1")]
    assert not any(t.startswith(("flat_store", "buffer_store", "scratch_")) for t in before)
    lastStore = max(i for i, t in enumerate(before) if t.startswith("global_store"))
    assert any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in before[lastStore + 1:]), "stamps not drained in front of the barrier"
    after = [t for t in behind if t.startswith(("buffer_wbl2", "s_waitcnt", "global_atomic", "global_store", "s_endpgm", "buffer_inv"))]
    seq = iter(after)

    def expect(pred, what):
        for t in seq:
            if pred(t):
                return t
            assert not t.startswith(("global_atomic", "global_store", "buffer_wbl2")), f"{what}: found {t} first"
        raise AssertionError(f"{what}: missing")

    wait = lambda t: t.startswith("s_waitcnt") and "vmcnt(0)" in t
    expect(lambda t: t.startswith("buffer_wbl2") and "sc1" in t, "write-back in front of the ticket")
    expect(wait, "wait behind the first write-back")
    atom = expect(lambda t: t.startswith("global_atomic_inc"), "ticket")
    assert "sc0" in atom                                           # a RETURNING atomic: the last ticket is told by its value
    expect(wait, "wait for the ticket's value")
    expect(lambda t: t.startswith("buffer_wbl2") and "sc1" in t, "write-back in front of the flag")
    expect(wait, "wait behind the second write-back")
    flag = expect(lambda t: t.startswith("global_store_dword "), "flag store")
    assert flag.endswith("sc1")
    # consumer side: in front of the first barrier, the first two global loads carrying sc1 are the poll and the word; the poll is
    # waited for before the word is asked for
    sc1 = [i for i, t in enumerate(head) if t.startswith("global_load") and t.endswith("sc1")]
    assert len(sc1) >= 2, "the consumer's poll and word loads must be sc1"
    assert any(wait(t) for t in head[sc1[0] + 1:sc1[1]])


def test_cull_desc_handoff(asm):
    """k_cull_desc's partial sums (root-culled edge): sc1 stores, fence, returning ticket; the last block reads with sc1."""
    ins = body(asm, "11k_cull_desc")
    assert any(t.startswith("global_atomic_add") and "sc0" in t for t in ins)
    assert any(t.startswith("buffer_wbl2") for t in ins)


def test_tie_update_of_the_closest_hit_kernel_has_no_lane_mask_operand():
    """docs/LAB_NOTES.md ("A MISCOMPILE FOUND ON THE WAY"): under ROCm 7.2 the form `tHit < best || (hit && tHit == best && f())`
    inside the divergent loop lost a lane's update.  The source must keep the three-plain-booleans form without `hit` in the tie
    (tHit == best implies a hit, best starts at 1e30 and tHit < 1e30); the pixel it broke is pinned by
    test_gpu_parity.py::test_closest_hit_mode_equals_the_earlier_shader[calgary] (the whole golden frame, bitwise)."""
    src = open(os.path.join(CSRC, "rto_device.hip.h")).read()
    k = src[src.index("void k_closest_near_first("):]
    k = re.sub(r"//[^\n]*", "", k[:k.index("\n}\n")])              # code only: the comment there quotes the bad form
    m = re.search(r"const bool tie = ([^;]+);", k)
    assert m, "the tie test must stay a named boolean"
    assert "hit" not in re.sub(r"tHit", "", m.group(1)), m.group(1)
    assert not re.search(r"\|\|\s*\([^)]*&&[^)]*pops_before", k), "short-circuit form of the tie update is back"


def test_no_short_circuit_updates_with_calls_inside_divergent_loops():
    """The shape that was miscompiled -- `a || (b && c && f(..))` deciding an update -- must not reappear in device code."""
    for name in ("rto_device.hip.h",):
        src = open(os.path.join(CSRC, name)).read()
        src = re.sub(r"//[^\n]*", "", src)
        bad = re.findall(r"if\s*\([^;{}]*\|\|\s*\([^;{}()]*&&[^;{}()]*&&[^;{}]*\w+\([^;{}]*\)\s*\)\s*\)\s*\{?[^;]*=", src)
        assert not bad, bad[:3]


def test_rccl_loopback_stand_in_builds_and_exports_what_rto_comm_binds(tmp_path):
    """tests/rccl_shim (test infrastructure of test_gpu_parity.py::test_comm_ranks_exchange_through_a_loopback_transport) must offer
    every symbol rto_comm.inc looks up in librccl -- read from the RTO_SYM(...) / dlsym lines themselves, so a new binding there turns
    this red here instead of the GPU test failing in its child process."""
    hipcc = _hipcc()
    if not hipcc:
        pytest.skip("no hipcc in this environment")
    so = tmp_path / "librccl.so.1"
    subprocess.run([hipcc, "-O1", "-shared", "-fPIC", "-std=c++17", os.path.join(ROOT, "tests", "rccl_shim", "rccl_shim.cpp"), "-o", str(so)],
                   check=True, stderr=subprocess.DEVNULL)
    exported = set(re.findall(r" T (nccl\w+)", subprocess.run(["nm", "-D", str(so)], check=True, capture_output=True, text=True).stdout))
    src = open(os.path.join(CSRC, "rto_comm.inc")).read()
    wanted = set(re.findall(r'"(nccl[A-Z]\w+)"', src))
    assert len(wanted) >= 12, wanted
    assert wanted <= exported, wanted - exported
