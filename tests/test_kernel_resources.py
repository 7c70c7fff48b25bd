"""Build-time pin of the hot kernels' register / scratch usage.

Round 4 removed the spills of the scan's epilogue and of the writer with opaque lane indices and a hand-placed `s_waitcnt`
(DESIGN 4.1, 4.2); those tricks depend on how the compiler happens to schedule, and the build container (ROCm 7.2) is not the only
toolchain this source will ever see.  The numbers below are read from the code objects of the library that was just built
(`pathfinder.jl_amd/tools/kernel_resources.py`: AMDGPU metadata notes of the gfx950 bundle entries); a toolchain or source change that
re-introduces scratch traffic or drops a kernel below the occupancy it was tuned for fails here, on the CPU, before anybody measures.
Bounds = the shipped build's values plus a few registers of slack; scratch bounds are what the kernel is KNOWN to tolerate
(`profiles/r04_experiments.md`: 56 B per thread in the scan's per-batch epilogue is outside the block loop).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathfinder.jl_amd", "tools"))

# kernel (demangled prefix) -> (max VGPRs incl. AGPRs, max scratch bytes per work-item, why)
PINS = {
    "pf_elbo_qf_kernel<12, 1, 8, 2>(": (256, 64, "config 3/4 scan: 2 waves per SIMD; only the per-batch epilogue may spill (round 4: 140 B -> 56 B)"),
    "pf_elbo_qf_kernel<12, 1, 0, 2>(": (256, 0, "config 2 scan (diagonal Gaussian)"),
    "pf_elbo_qf_kernel<20, 2, 0, 2>(": (256, 64, "config 5 scan (funnel, J = 10): register-lean body of round 3, no spill inside the block loop"),
    "pf_elbo_xw_kernel<12>(": (216, 0, "draw writer at J = 6: 207 - 209 VGPRs, no scratch (round 4: opaque lane indices)"),
    "pf_elbo_xw_kernel<20>(": (256, 128, "draw writer at J = 10 (round 4: 264 -> 96-104 B of scratch)"),
    "pf_fit_reg_kernel<12, 4, 256>(": (168, 0, "fit at d <= 1024, J = 6: 3 workgroups per CU need <= 168 VGPRs, no spill (round 4: 163)"),
    "pf_fit_reg_kernel<12, 2, 256>(": (128, 0, "fit at d <= 512"),
    "pf_fit_tsqr_kernel<20, 3>(": (256, 512, "large-d fit at J = 10 (round 6): 3 rows x 20 columns per thread in registers; the stage functions' few spills, nothing in the column loop"),
    "pf_fit_tsqr_kernel<12, 5>(": (256, 512, "large-d fit at J = 6"),
    "pf_history_kernel<4, 256>(": (256, 0, "history walk at d <= 1024: four register sets of rows, nothing spilled"),
    "pf_lbfgs_kernel<4, 256, 8, true>(": (512, 0, "device L-BFGS at config 3 (rank-8 target, ring in LDS): VGPRs + AGPRs at one wave per SIMD, no scratch"),
    "pf_lbfgs_kernel<4, 256, 0, true>(": (512, 0, "device L-BFGS, diagonal / funnel target"),
}


@pytest.fixture(scope="module")
def table():
    import kernel_resources as kr
    import pfmi
    pfmi.build()
    t = kr.kernel_resources()
    assert len(t) > 300, len(t)                       # every translation unit's bundle was found and parsed
    return t


@pytest.mark.parametrize("prefix", sorted(PINS))
def test_hot_kernel_register_and_scratch_pins(table, prefix):
    vmax, smax, why = PINS[prefix]
    hits = [k for k in table if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, hits)
    r = table[hits[0]]
    regs = r["vgpr_count"]                            # (.vgpr_count is the unified file: it already includes the AGPR block on gfx950)
    assert regs <= vmax, f"{hits[0]}: {regs} VGPRs > {vmax} ({why})"
    assert r["private_segment_fixed_size"] <= smax, f"{hits[0]}: {r['private_segment_fixed_size']} B scratch per work-item > {smax} ({why})"
    if smax == 0:
        assert r["vgpr_spill_count"] == 0, (hits[0], r)


def test_no_kernel_uses_dynamic_stack_or_huge_scratch(table):
    """nothing in the library may fall off a cliff: > 4 KB of scratch per work-item means a register array went to memory"""
    # (the kernels of the history_length 17 .. 32 route -- column padding 64 -- are the general kernels instantiated beyond what registers hold:
    #  their per-draw vectors live in scratch BY DESIGN, the route is documented as slow-but-correct; a real cliff elsewhere still fails)
    bad = {k: v["private_segment_fixed_size"] for k, v in table.items()
           if v.get("private_segment_fixed_size", 0) > 4096 and "<64" not in k.split("(")[0]}
    assert not bad, bad
    big = {k: v["private_segment_fixed_size"] for k, v in table.items() if "<64" in k.split("(")[0]}
    assert big and max(big.values()) <= 20480, big
