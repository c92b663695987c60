"""Properties of the generated gfx950 code that the scan kernel's hand-placed instructions rely on (CPU tier: hipcc
cross-compiles without a GPU).  k_scan_reads loads its vectors with global_load_lds_dwordx4 from inline assembly: the
assembly sets M0 (a register the compiler reserves) and waits for the loads by hand, so nothing else in the kernel may
touch M0, and the kernel must keep the register / LDS budget its launch shape assumes (2 blocks of 8 waves per CU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def scan_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "extract.s")
    src = os.path.join(ROOT, "badger_amd", "csrc", "extract_kernels.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-function",
                    "-Wno-inline-asm", "-Wno-unused-command-line-argument", "-o", out, src], check=True, timeout=600)
    text = open(out).read()
    m = re.search(r"^(_ZN\S*k_scan_reads\S*):[^\n]*\n(.*?)\n\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, "k_scan_reads not found in the generated code"
    meta = dict(re.findall(r"\.set \S*k_scan_reads\S*\.(num_vgpr|private_seg_size|numbered_sgpr), (\d+)", text))
    return m.group(2).split("\n"), m.group(3), meta, text


def test_m0_is_written_only_for_the_lds_dma(scan_isa):
    body, _, _, _ = scan_isa
    code = [l.strip() for l in body if l.strip() and not l.strip().startswith(";")]
    uses = [i for i, l in enumerate(code) if re.search(r"\bm0\b", l)]
    dmas = [i for i, l in enumerate(code) if l.startswith("global_load_lds_dwordx4")]
    assert dmas, "the LDS-DMA loads are gone"
    for i in uses:
        assert code[i].startswith("s_mov_b32 m0,"), "M0 touched outside the DMA sequence: " + code[i]
        assert any(0 < d - i <= 3 for d in dmas), "an M0 write that is not followed by its load"
    assert len(uses) == len(dmas)
    # every wait on the vector-memory counter inside the step loop is one of the hand-placed ones or a full drain
    assert any("s_waitcnt vmcnt(1)" in l for l in code)


def test_scan_kernel_budget(scan_isa):
    _, desc, meta, _ = scan_isa
    assert int(meta["private_seg_size"]) == 0, "k_scan_reads spills"
    assert int(meta["num_vgpr"]) <= 128, "more than 128 registers: fewer than 4 waves per SIMD"
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", desc).group(1))
    assert 2 * lds <= 160 * 1024, "two blocks no longer fit a CU's LDS"


def test_alignment_kernel_budget(scan_isa):
    """k_sw_clusters (round 4): the three-way maximum of a cell is the packed half-float instruction, the match bonus a packed
    multiply-add, the windows live in LDS - 96 registers where the integer form with the windows in registers took 240.
    Five blocks of four waves per CU (30 KB of LDS a block) is what the kernel is sized for."""
    text = scan_isa[3]
    m = re.search(r"^(_ZN\S*k_sw_clusters\S*):[^\n]*\n(.*?)\n\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, "k_sw_clusters not found in the generated code"
    body, desc = m.group(2), m.group(3)
    meta = dict(re.findall(r"\.set \S*k_sw_clusters\S*\.(num_vgpr|private_seg_size), (\d+)", text))
    assert int(meta["private_seg_size"]) == 0, "k_sw_clusters spills"
    assert int(meta["num_vgpr"]) <= 96, "more than 96 registers: fewer than 5 waves per SIMD"
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", desc).group(1))
    assert 5 * lds <= 160 * 1024, "five blocks no longer fit a CU's LDS"
    assert body.count("v_pk_maximum3_f16") >= 2 * 33 * 4, "the three-way maxima are not single instructions any more"
    assert body.count("v_pk_mad_u16") >= 2 * 22 * 4, "the match bonus is not a multiply-add any more"


def test_deletion_variant_join_kernels_budget(tmp_path):
    """The deletion-variant join's kernels are sized by hand from these: no scratch, native LDS minimum for the row's
    repeat table, and an LDS footprint that lets several blocks share a compute unit (the row passes run 8 blocks of 4
    waves, the pair kernel 3 blocks of 8 waves, the bucket split one block of 16).  No sort library is left on the path."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "graph.s")
    src = os.path.join(ROOT, "badger_amd", "csrc", "graph_kernels.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-function",
                    "-Wno-inline-asm", "-Wno-unused-command-line-argument", "-I", os.path.join(ROOT, "include"), "-o", out, src],
                   check=True, timeout=900)
    text = open(out).read()
    for name, max_vgpr, max_lds in (("k_d2_rowsILb0ELj1024", 64, 20 * 1024), ("k_d2_rowsILb1ELj1024", 64, 8 * 1024), ("k_d1_rowsILb1ELj1024", 64, 4 * 1024),
                                    ("k_part_splitIj", 128, 100 * 1024), ("k_d2_pairs_wILi2", 128, 40 * 1024), ("k_d2_pairs_wILi1", 128, 40 * 1024),
                                    ("k_d2_pairsILi2", 128, 53 * 1024 + 512), ("k_d2_pairsILi1", 128, 53 * 1024 + 512)):
        m = re.search(r"^(_ZN\S*%s\S*):[^\n]*\n(.*?)\n\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel" % name, text, re.S | re.M)
        assert m, name + " not found in the generated code"
        meta = dict(re.findall(r"\.set \S*%s\S*\.(num_vgpr|private_seg_size), (\d+)" % name, text))
        assert int(meta["private_seg_size"]) == 0, name + " spills"
        assert int(meta["num_vgpr"]) <= max_vgpr, (name, meta["num_vgpr"])
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", m.group(3)).group(1))
        assert lds <= max_lds, (name, lds)
        if name.startswith("k_d2_rowsILb0"):              # (the counting run settles a row's repeats; the second run reads its masks)
            assert "ds_min_u32" in m.group(2) and "ds_cmpst" not in m.group(2), name + ": the table update is no longer one LDS instruction"
