"""GPU: the parts of the C ABI a non-Python caller relies on — structured GAFStruct fields (no text parsing), the
multi-device call, handle re-use with new reads."""
import pytest

pytestmark = pytest.mark.gpu

MODES = ("MODE_GLOBAL_POA", "MODE_GLOBAL_POA_SCALAR", "MODE_GAP_POA", "MODE_LOCAL_POA", "MODE_GAP_LOCAL_POA", "MODE_PATHWISE",
         "MODE_PATHWISE_SEMI", "MODE_RECOMBINATION", "MODE_RECOMBINATION_SEMI")


def test_result_fields_rebuild_the_gaf_line(example_gfa, example_reads):
    """rg_result_fields carries every field of the reference's GAFStruct (gaf_output.rs:6-20): GAFStruct::to_string
    over them is the record line rg_result_gaf prints, in every mode, including the empty record of a band failure."""
    from recgraph_amd import api, synth
    names, reads = example_reads
    g = api.Graph.from_gfa_text(example_gfa)
    seen_empty = seen_warning = False
    for mname in MODES:
        b = api.Batch(g, reads, api.make_params(getattr(api, mname)))
        b.run()
        b.fetch()
        for i in range(len(reads)):
            text = b.gaf_text(i, names[i], i + 1)
            f = b.fields(i, names[i])
            assert f is not None and f.to_string() + "\n" == text.split("\n", text.count("\n") - 1)[-1], (mname, i)
            seen_empty |= f.to_string() == api.GAFStruct().to_string()
    # a band failure (GAFStruct::new()) and a band warning, on a long linear-ish graph with short reads
    sg = synth.linear_graph(1500, seed=9)
    rd = synth.substring_reads(sg, 60, 80, seed=10)
    g2 = api.Graph.from_gfa_text(sg.gfa())
    for mode in (api.MODE_GLOBAL_POA, api.MODE_GAP_POA):
        b = api.Batch(g2, rd, api.make_params(mode, b=1.0, f=0.0))
        b.run()
        b.fetch()
        for i in range(len(rd)):
            text = b.gaf_text(i, "q", i + 1)
            f = b.fields(i, "q")
            lines = text.split("\n")[:-1]
            assert f.to_string() == lines[-1]
            seen_empty |= f.to_string() == api.GAFStruct().to_string()
            seen_warning |= len(lines) > 1
    assert seen_empty and seen_warning


def test_multi_device_call_equals_one_batch(oracle):
    """rg_align_batch_multi: the streaming engine behind one call — tiles of contiguous reads (at least one per entry of
    device_ids) pulled by the batch handles of every listed device, text in input order.  One GPU here: the device list
    names it several times (its handles then share the device and its graph tables)."""
    from recgraph_amd import api, synth
    sg = synth.haplotype_graph(1500, 8, path_len=300, seed=11)
    reads = synth.haplotype_reads(sg, 37, length=300, seed=12, mosaic_frac=0.5)
    names = ["q%d" % i for i in range(len(reads))]
    g = api.Graph.from_gfa_text(sg.gfa())
    for mode in (api.MODE_RECOMBINATION, api.MODE_PATHWISE, api.MODE_GLOBAL_POA, api.MODE_GAP_POA):
        one, _ = api.align_batch(g, reads, names, mode=mode)
        for devs in ([0], [0, 0, 0], None):
            m = api.MultiBatch(g, reads, api.make_params(mode), device_ids=devs)
            assert m.begin[0] == 0 and m.begin[-1] == len(reads) and len(m.shards) >= (len(devs) if devs else 1)
            assert m.begin == sorted(m.begin) and sum(sh.n for sh in m.shards) == len(reads)
            assert m.format_all(names, 1, 4).decode() == "".join(one), (mode, devs)
            sh, j = m.locate(20)
            assert sh.gaf_text(j, names[20], 21) == one[20]
    og = oracle.Graph.from_gfa_text(sg.gfa())
    assert one[5] == og.align(oracle.M2, reads[5], name=names[5], idx=6)[0]
    with pytest.raises(Exception):
        api.MultiBatch(g, reads, api.make_params(api.MODE_PATHWISE), device_ids=[0, 99])    # no such device: status code


def test_handle_reuse_with_new_reads(oracle):
    """rg_batch_set_reads: one handle, several read sets of different sizes and lengths; same records as fresh handles."""
    from recgraph_amd import api, synth
    sg = synth.haplotype_graph(1500, 8, path_len=300, seed=11)
    g = api.Graph.from_gfa_text(sg.gfa())
    sets = [synth.haplotype_reads(sg, n, length=ln, seed=s, mosaic_frac=0.5) for n, ln, s in ((20, 300, 1), (45, 120, 2), (7, 330, 3), (20, 300, 1))]
    for mode in (api.MODE_RECOMBINATION, api.MODE_GLOBAL_POA, api.MODE_GAP_POA, api.MODE_LOCAL_POA):
        b = api.Batch(g, sets[0], api.make_params(mode))
        for k, rs in enumerate(sets):
            if k:
                b.set_reads(rs)
            b.run()
            b.fetch()
            fresh, _ = api.align_batch(g, rs, None, mode=mode)
            assert [b.gaf_text(i, "read%d" % i, i + 1) for i in range(len(rs))] == fresh, (mode, k)
