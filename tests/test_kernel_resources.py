"""CPU: the headline variants of k_sweep16 use no scratch.  hipcc cross-compiles gfx950 without a GPU;
`-Rpass-analysis=kernel-resource-usage` reports registers, spills and scratch per kernel (tools/kernel_resources.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_headline_sweep_variants_use_no_scratch():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    ks = {k["name"]: k for k in kernel_resources.report("rg_sweep16.hip")}
    # (template arguments: C, kColmax, kRec, kWide, kSemi)
    headline = ["rg::k_sweep16<16, 0, true, false, false>",     # -m 8, both sweeps since round 5: records, no column maxima in the sweep
                "rg::k_sweep16<16, 0, false, false, false>"]    # -m 4: no tracking at all
    for name in headline:
        k = ks[name]
        assert k["ScratchSize [bytes/lane]"] == 0 and k["VGPRs Spill"] == 0, (name, k)
        assert k["VGPRs"] <= 256 and k["Occupancy [waves/SIMD]"] >= 2, (name, k)
    # the -m 4 sweep fits three waves per SIMD (168 registers, 13 KB of LDS per wave)
    assert ks["rg::k_sweep16<16, 0, false, false, false>"]["Occupancy [waves/SIMD]"] >= 3
    # reads of 1024-2047 bases (32 columns per lane): a row is 16 registers and the record variant is at the 256-register limit —
    # a few spilled registers are tolerated, a relapse (one uniform branch in the alpha took it from 13 to 67) is not
    k32 = ks["rg::k_sweep16<32, 0, true, false, false>"]
    assert k32["VGPRs Spill"] <= 24 and k32["ScratchSize [bytes/lane]"] <= 96, k32
    # the narrower instantiations of the same variants (shorter reads) and their semiglobal forms do not spill either
    for c in (4, 8):
        for v in ("0, true, false, false", "0, false, false, false", "0, true, false, true", "0, false, false, true"):
            k = ks["rg::k_sweep16<%d, %s>" % (c, v)]
            assert k["ScratchSize [bytes/lane]"] == 0, (c, v, k)


def test_no_valu_write_into_a_wide_buffer_store_in_flight():
    """The gfx950 hazard of round 5 (rg_sweep16.hip, st_row): no buffer_store_dwordx3/x4 of any k_sweep16 variant has its
    data registers overwritten by the instruction behind it — and the scan itself still finds the pattern when the s_nop
    is compiled out."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    hits, stores = kernel_resources.store_hazards("rg_sweep16.hip")
    assert stores > 100 and not hits, hits[:4]
    # (the other sources use no buffer intrinsics: the compiler's own wide stores are FLAT / scratch, whose wait state it inserts)
    csrc = os.path.join(ROOT, "recgraph_amd", "csrc")
    users = [f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".cpp")) and "raw_buffer_store" in open(os.path.join(csrc, f)).read()]
    assert users == ["rg_sweep16.hip"], users
    hits, _ = kernel_resources.store_hazards("rg_sweep16.hip", ["-DRG_SWEEP16_NO_STORE_NOP"])
    assert any("k_sweep16<16, 0, false, false, false>" in h[0] for h in hits), hits[:4]


def test_poa_kernels_use_no_scratch():
    """`k_m0_simd` and `k_poa_banded` keep the previous row's chunks in registers — the compiler once fused the select chains
    that pick a chunk into a dynamically indexed vector it kept in scratch (32 / 48 bytes per lane, dependent scratch loads in
    the fast path of every row; profiles/r04_notes.md).  No variant of either may allocate scratch."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    seen = 0
    for src, prefix in (("rg_poa.hip", "rg::k_m0_simd<"), ("rg_poa_banded.hip", "rg::k_poa_banded<")):
        for k in kernel_resources.report(src):
            if k["name"].startswith(prefix):
                seen += 1
                assert k["ScratchSize [bytes/lane]"] == 0 and k["VGPRs Spill"] == 0, k
    assert seen == 8
