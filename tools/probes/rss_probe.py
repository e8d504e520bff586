#!/usr/bin/env python3
"""Where the resident memory of a streaming process comes from (VERDICT r3 #5 asked for a CLI below 300 MB): RSS after
each stage of a bounded config-5 stream.  python3 tools/probes/rss_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def rss():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"):
            return round(int(ln.split()[1]) / 1024.0, 1)


out = {"python": rss()}
from recgraph_amd import _lib, api, synth  # noqa: E402
out["numpy + ctypes binding"] = rss()
_lib.load()
out["librecgraph_hip.so loaded (HIP runtime mapped)"] = rss()
_lib.load().rg_device_count()
out["hipGetDeviceCount"] = rss()
sg, _, _ = synth.make_config("C5", n_reads=1)
g = api.Graph.from_gfa_text(sg.gfa())
out["graph"] = rss()
st = api.Stream(g, api.make_params(8), device_ids=[0], max_queued_tiles=4, max_undelivered_bytes=64 << 20)
out["rg_stream_create (3 worker threads, device context)"] = rss()
reads = api.Batch.pack_reads(synth.haplotype_reads(sg, 4096, 1000, seed=1, mosaic_frac=0.5))
out["one packed tile of reads in Python"] = rss()
for k in range(3):
    st.push(reads)
for k in range(3):
    st.next()
out["after 3 tiles (3 handles exist: pinned staging, host records)"] = rss()
for k in range(12):
    st.push(reads)
    st.next()
out["after 15 tiles"] = rss()
st.finish()
st.close()
print(json.dumps(out))
