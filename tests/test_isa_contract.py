"""Properties of the generated gfx950 ISA that DESIGN.md relies on (register budget / occupancy of
the hot kernels, 16-byte state accesses, LDS-staged observation rows, no barrier, no MFMA), checked
from a cross-compile so that a compiler or source change cannot silently lose them.  CPU only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("isa")
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import HIPCC_FLAGS, PARTS
    # the library's translation units (MDS_PART), compiled to assembly side by side like build() compiles them to objects
    procs = [subprocess.Popen(["hipcc", *HIPCC_FLAGS, f"-DMDS_PART={k}", "-S", "--cuda-device-only", "-o", str(d / f"mds{k}.s"),
                               os.path.join(ROOT, "multidronesim_amd", "csrc", "mds_api.hip")], stderr=subprocess.DEVNULL) for k in PARTS]
    assert all(p.wait() == 0 for p in procs)
    return "\n".join(open(d / f"mds{k}.s").read() for k in PARTS)


def kernel_ops(isa, mangled):
    i = isa.index("\n" + mangled + ":")
    j = isa.index("s_endpgm", i)
    lines = [l.strip() for l in isa[i:j].split("\n")]
    return [l for l in lines if l and not l.startswith((";", "//", ".")) and not l.endswith(":")]


def test_hot_kernels_have_no_scratch_and_expected_occupancy(isa):
    for name, max_vgpr in (("_ZN3mds16k_step_geometricIffLb1ELb0ELb0ELb0ELb0EEEvNS_6ConstsIT_EEimdPT0_PKS2_PS2_S5_S5_iS5_", 64),
                           ("_ZN3mds6k_stepIffLb1ELb0ELb0ELb0EEEvNS_6ConstsIT_EEimPT0_PKS2_PS2_PKS4_S5_iS5_", 64)):
        meta = isa[isa.index("amdhsa.kernels:"):]
        blk = next(b for b in meta.split("\n  - ") if re.search(r"\.name:\s+" + re.escape(name) + r"\n", b))
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        sc = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        assert sc == 0, (name, sc)
        assert vg <= max_vgpr, (name, vg)     # 8 waves/SIMD


def test_no_mfma_and_no_barrier_in_hot_kernel(isa):
    ops = kernel_ops(isa, "_ZN3mds16k_step_geometricIffLb1ELb0ELb0ELb0ELb0EEEvNS_6ConstsIT_EEimdPT0_PKS2_PS2_S5_S5_iS5_")
    assert not [o for o in ops if "mfma" in o]
    assert not [o for o in ops if o.startswith("s_barrier")]          # wave-scope LDS staging only
    # obs span 5 x 1 KiB per wave (a full wave: all five chunks read from LDS, then stored; the shard's last, partial wave: chunk by chunk) + 3 packed state groups
    assert len([o for o in ops if o.startswith("global_store_dwordx4")]) == 5 + 5 + 3
    # the observation rows are a write-once stream: every one of their stores carries the non-temporal hint (a run-time policy flag once cost
    # it -- two stores in the arms of one branch are merged into a plain one: form 1 15.6 -> 16.9 us, C5 8.4 -> 9.7)
    assert len([o for o in ops if o.startswith("global_store_dwordx4") and o.endswith(" nt")]) == 10
    assert len([o for o in ops if o.startswith("ds_write_b128")]) == 5
    assert len([o for o in ops if o.startswith("global_load_dwordx4")]) == 4        # 3 state groups + (a, omega, yaw_rate, phase)
    assert len([o for o in ops if o.startswith("global_load_dword ") or o.startswith("global_load_dword\t")]) <= 2


def test_persistent_cbf_kernel_keeps_its_step_loop_free_of_scratch(isa):
    """k_cbf_rollout<float, geometric nominal>: 4 096 wavefronts = 4 per SIMD for C4, so it must fit 128 VGPRs; and nothing may be
    spilled inside the step loop -- a dozen serial scratch reloads per step cost a quarter of the kernel's time in round 3 (DESIGN.md
    section 4, C4 (e)).  Spill slots outside the loop (the launch prologue) are tolerated; none is expected today."""
    name = "_ZN3mds13k_cbf_rolloutIfLi0ELb0ELi8ELb0EEEvNS_8RollArgsIT_EE"       # (PAD = false: the C4 shape, D = 16)
    meta = isa[isa.index("amdhsa.kernels:"):]
    blk = next(b for b in meta.split("\n  - ") if re.search(r"\.name:\s+" + re.escape(name) + r"\n", b))
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 128
    assert int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1)) <= 80 * 1024       # two workgroups per CU
    ops = kernel_ops(isa, name)
    bars = [k for k, o in enumerate(ops) if o.startswith("s_barrier")]
    assert len(bars) == 3                                            # one per launch (the obstacle table before the first stage A), then before and after stage B
    bars = bars[1:]
    loop = ops[bars[0]:]                                             # (stage C + A follow the second barrier up to the loop's back edge)
    assert not [o for o in loop if o.startswith("scratch_")], [o for o in loop if o.startswith("scratch_")][:4]
    # per-lane planes are addressed as scalar base + 32-bit VGPR offset: no global access of the stage forms a 64-bit per-lane address
    # in the step loop except the last-step-only stores of the state / RPM planes
    late = ops[bars[1]:]
    assert len([o for o in late if o.startswith("global_load") and ", off" in o and "s[" not in o]) <= 2


def test_compare_models_kernel_streams_through_lds_with_full_occupancy(isa):
    """k_compare_models<float, float> (simulations/CompareModels.py:48-56 as one launch): HBM-bound streaming -- 8 waves per SIMD
    (<= 64 VGPRs, 20 KiB of LDS per 256-thread workgroup = 8 workgroups per CU), no scratch, no workgroup barrier (wave-scope staging),
    rows in and out of memory as 16-byte chunks only, the dense 12 x 12 / 12 x 4 model matrices read as scalar operands (s_load), no MFMA."""
    name = next(m.group(1) for m in re.finditer(r"\.name:\s+(_ZN3mds16k_compare_modelsIffEE\S+)", isa))
    meta = isa[isa.index("amdhsa.kernels:"):]
    blk = next(b for b in meta.split("\n  - ") if re.search(r"\.name:\s+" + re.escape(name) + r"\n", b))
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 64
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0
    assert int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1)) == 256 * 20 * 4
    ops = kernel_ops(isa, name)
    assert not [o for o in ops if "mfma" in o or o.startswith("s_barrier") or o.startswith("scratch_")]
    mem = [o for o in ops if o.startswith(("global_load", "global_store"))]
    assert mem and all(o.startswith(("global_load_dwordx4", "global_store_dwordx4")) for o in mem), mem
    assert len([o for o in ops if o.startswith("s_load_dwordx")]) >= 12          # 193 matrix / constant words through the scalar path
    assert len([o for o in ops if o.startswith("ds_read_b128")]) >= 5 and len([o for o in ops if o.startswith("ds_write_b128")]) >= 5
