"""The LDS-DMA variant of the fused kernel waits with a hand-counted ``s_waitcnt vmcnt(N)``
(the compiler does not track LDS-DMA completion).  N must equal the number of vector-memory
instructions a wave issues between starting the next batch's DMA and that wait.  This test
re-derives N from the generated gfx950 ISA so a compiler or source change cannot silently
break the count.  CPU only (hipcc cross-compiles)."""
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
    out = tmp_path_factory.mktemp("isa") / "mds.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                           os.path.join(ROOT, "multidronesim_amd", "csrc", "mds_api.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_ops(isa, mangled):
    i = isa.index("\n" + mangled + ":")
    j = isa.index("s_endpgm", i)
    lines = [l.strip() for l in isa[i:j].split("\n")]
    return [l for l in lines if l and not l.startswith((";", "//", ".")) and not l.endswith(":")]


@pytest.mark.parametrize("obs,act,expect", [(1, 0, 18), (1, 1, 19), (0, 0, 13), (0, 1, 14)])
def test_dma_kernel_store_count_matches_vmcnt(isa, obs, act, expect):
    name = f"_ZN3mds24k_step_geometric_f32_dmaILb{obs}ELb{act}ELb0EEEvNS_6ConstsIfEEimdPfPKfS3_S3_S3_"
    ops = kernel_ops(isa, name)
    stores = [o for o in ops if o.startswith("global_store")]
    dmas = [o for o in ops if o.startswith("global_load_lds_dword")]
    plain_loads = [o for o in ops if o.startswith("global_load") and not o.startswith("global_load_lds")]
    assert len(stores) == expect, stores
    assert len(dmas) == 40                      # 20 rows for the first batch + 20 in the loop body
    assert not plain_loads
    waits = [o for o in ops if o.startswith("s_waitcnt") and "vmcnt" in o]
    assert any(f"vmcnt({expect})" in w for w in waits), waits
    # the only full drain is the one in front of the loop (first batch)
    assert sum("vmcnt(0)" in w for w in waits) == 1, waits
    # slice reads are the explicit ds_read_b32 with 256-byte row offsets, drained by one lgkmcnt(0)
    reads = [o for o in ops if o.startswith("ds_read_b32")]
    assert len(reads) == 20
    offs = sorted(int(re.search(r"offset:(0x[0-9a-f]+|\d+)", r).group(1), 0) if "offset" in r else 0 for r in reads)
    assert offs == [256 * p for p in range(20)]


def test_hot_kernels_have_no_scratch_and_expected_occupancy(isa):
    for name, max_vgpr in (("_ZN3mds16k_step_geometricIffLb1ELb0ELb0ELb0EEEvNS_6ConstsIT_EEimdPT0_PKS2_PS2_S5_S5_", 80),
                           ("_ZN3mds6k_stepIffLb1ELb0ELb0EEEvNS_6ConstsIT_EEimPT0_PKS2_PS2_PKS4_S5_", 80)):
        meta = isa[isa.index("amdhsa.kernels:"):]
        blk = next(b for b in meta.split("\n  - ") if re.search(r"\.name:\s+" + re.escape(name) + r"\n", b))
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        sc = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        assert sc == 0, (name, sc)
        assert vg <= max_vgpr, (name, vg)     # >= 6 waves/SIMD


def test_no_mfma_and_no_barrier_in_hot_kernel(isa):
    ops = kernel_ops(isa, "_ZN3mds16k_step_geometricIffLb1ELb0ELb0ELb0EEEvNS_6ConstsIT_EEimdPT0_PKS2_PS2_S5_S5_")
    assert not [o for o in ops if "mfma" in o]
    assert not [o for o in ops if o.startswith("s_barrier")]          # wave-scope LDS staging only
    assert len([o for o in ops if o.startswith("global_store_dwordx4")]) == 5   # obs span: 5 x 1 KiB per wave
    assert len([o for o in ops if o.startswith("ds_write_b128")]) == 5
