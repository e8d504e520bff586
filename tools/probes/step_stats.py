"""Shape of a configuration's step tables (no GPU): how the member-row updates of one sweep split over register runs (by
group size), gather runs, tails and general records — the static weights for tools/kernel_resources.py --isa.
    python tools/probes/step_stats.py [C5] [fwd|rev] [plain|split]"""
import collections
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recgraph_amd import api, synth      # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
fwd = (sys.argv[2] if len(sys.argv) > 2 else "fwd") == "fwd"
split = (sys.argv[3] if len(sys.argv) > 3 else "split") == "split"
g, reads, _ = synth.make_config(cfg, n_reads=4)[:3]
gg = api.Graph.from_gfa_text(g.gfa())
which = (31 if fwd else 33) + (1 if split else 0)
text = gg.dump(which)
head, recs, lead = text.split(";")[0:2], text.split(";")[2], None
recs = [tuple(int(x, 16) for x in r.split(":")) for r in recs.split(",") if r]
KRUN = 4
kinds = collections.Counter()
members = collections.Counter()
t = 0
while t < len(recs):
    x, y, z, w = recs[t]
    f = (x >> 23) & 7
    field = (x >> 26) & 63
    nm = bin(z).count("1") + bin(w).count("1")
    if (f & 4) and field != 0:
        # head (4) or inner (7): a run of `field` rows (capped at 63)
        R = field
        if nm <= KRUN:
            kind = "regrun_nm%d" % nm
        elif R * (84 * (nm - 1) - 160) >= 90 * (nm - 1):          # (RG_GATHER_PER_* of rg_sweep16.hip)
            kind = "gather"
        else:
            kind = "general_inner"
        kinds[kind + "_rows"] += 1
        members[kind] += nm
        if kind == "gather":
            kinds["gather_nm_%02d" % (nm // 4 * 4)] += 1
    elif (f & 4):
        kinds["tail_nm%d_rows" % nm] += 1
        members["tail"] += nm
    else:
        kinds["general_nm%s_rows" % (nm if nm < 5 else "5+")] += 1
        members["general"] += nm
    t += 1
tot = sum(members.values())
print(cfg, "fwd" if fwd else "rev", "split" if split else "plain", "records", len(recs), "member rows", tot)
for k in sorted(members):
    print("  member rows %-16s %7d  %.3f" % (k, members[k], members[k] / tot))
for k in sorted(kinds):
    print("  records     %-20s %7d" % (k, kinds[k]))
