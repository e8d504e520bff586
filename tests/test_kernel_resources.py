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
    # the -m 4 sweep: two waves per SIMD since the path retirement of round 6 went in (196 registers; compiled for three it spills
    # 25 and the stream loses: 279-319 k against 324-370 k reads/s at config 4) — and it must leave the small kernels their room
    assert ks["rg::k_sweep16<16, 0, false, false, false>"]["VGPRs"] <= 200
    # reads of 1024-2047 bases (32 columns per lane): a row is 16 registers and the record variant is at the 256-register limit —
    # a few spilled registers are tolerated, a relapse (one uniform branch in the alpha took it from 13 to 67) is not
    # (round 6: gather runs are compiled into this variant — 32 spilled registers, 132 bytes of scratch per lane — because they
    # pay all the same: 21.4 k -> 24.9 k reads/s at 1.5 kbp, profiles/r06_notes.md)
    k32 = ks["rg::k_sweep16<32, 0, true, false, false>"]
    assert k32["VGPRs Spill"] <= 40 and k32["ScratchSize [bytes/lane]"] <= 160, k32
    # the record variant leaves 64 of a SIMD's 512 registers to the other handles' small kernels (two waves of <= 224), and
    # k_layer16 at <= 16 columns per lane fits into them: one more allocation granule on either side costs 2-3 % in the stream
    assert ks["rg::k_sweep16<16, 0, true, false, false>"]["VGPRs"] <= 224, ks["rg::k_sweep16<16, 0, true, false, false>"]
    kl = next(k for n, k in ks.items() if n.startswith("rg::k_layer16<16>"))
    assert kl["VGPRs"] <= 64 and kl["VGPRs Spill"] <= 2, kl
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
    # round 6: the scan reads global / flat / SCRATCH stores too (register spills are scratch_store_dwordx4 with an SGPR offset) and
    # every destination operand; the other big kernel source — the i32 sweeps, hundreds of spilled registers — is clean as well
    hits2, stores2 = kernel_resources.store_hazards("rg_pathwise.hip")
    assert stores2 > 100 and not hits2, hits2[:4]
    # ... and the scan finds the pattern when it is there (round 5 checked this on the build without the s_nop; the round-6 tree
    # happens to schedule no such pair even then, so the positive control is a listing written by hand: the round-5 instruction
    # pair, a spill store whose register is reused at once, a v_swap's second operand — and three pairs that must NOT count)
    listing = """_Z3foov:
	buffer_store_dwordx4 v[126:129], v119, s[44:47], s8 offen
	v_and_b32_e32 v126, 1, v3
	buffer_store_dwordx4 v[10:13], v119, s[44:47], s8 offen
	s_nop 1
	v_and_b32_e32 v10, 1, v3
	scratch_store_dwordx4 off, v[20:23], s32 offset:16 ; 16-byte Folded Spill
	v_mov_b32_e32 v22, 0
	global_store_dwordx3 v[2:3], v[30:32], off
	v_add_u32_e32 v5, v30, v31
	global_store_dwordx4 v40, v[50:53], s[4:5]
	v_swap_b32 v9, v53
	scratch_store_dwordx2 off, v[60:61], off offset:8
	v_mov_b32_e32 v60, 0
""".split("\n")
    hits, stores = kernel_resources.scan_store_hazards(listing)
    assert stores == 5 and [h[1] for h in hits] == [2, 7, 11], hits
    assert len(kernel_resources.scan_store_hazards(listing, want64=True)[0]) == 4


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
