"""Dev helper: merge the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/pmc_pass.sh) into profiles/pmc_traffic.json
and a markdown table.  HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction of the guide)."""
import json, sys, os
d = sys.argv[1]; tag = sys.argv[2]; batch = sys.argv[3] if len(sys.argv) > 3 else '64'
f, w = {}, {}
for dd in d.split(','):          # one directory per engine (tools/round_profile.sh: fp16x3 and bf16 passes), merged
    f.update(json.load(open(os.path.join(dd, 'FETCH_SIZE', 'summary.json')))); w.update(json.load(open(os.path.join(dd, 'WRITE_SIZE', 'summary.json'))))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
prev = os.path.join(root, 'profiles', 'pmc_traffic.json')
if os.path.exists(prev):          # kernels not in these passes (the other engine's) keep their last measured value
    out.update({k: v for k, v in json.load(open(prev)).items() if k not in ('note', 'read_factor')})
out['note'] = f'per-launch HBM bytes from PMC at sub-batch {batch} (latest passes: profiles/{tag}_pmc_hbm_traffic.md; kernels absent from them keep their previous value)'
rows = []
def label(k):
    return ('headx3_fused' if 'headx3' in k else 'head_fused' if 'head32_kernel' in k or 'head_fused' in k else 'bblockx3_fused' if 'bblockx3' in k
            else 'bblock48_fused' if 'bblock48_kernel' in k else k)
# Read correction per kernel: FETCH_SIZE x 1 KB reports HALF of a wide coalesced read stream on gfx950 (x2, the guide's correction), but
# ALL of the bytes of isolated 64-byte pieces 1600 B apart and 3/4 of 128-byte pieces at that stride (tools/dev/fetch_probe.hip,
# NOTES/design_history_r1_r5.md §9.6 item 7).  The fused heads read their gather boxes in exactly such pieces (plus one coalesced stream, the stem tensor):
# the measured mix is ~1.3; every other kernel reads coalesced streams (LDS-DMA halo rows, weight fragments, fp32 rows).
# The stem's second convolution (64 -> 64, stride 2, fp32 rows of 256 B per pixel): x2 gives 7.23 GB per 865 us launch = 8.4 TB/s, more than the
# HBM can deliver -- for that stream the counter evidently reports the bytes in full; x1 (3.9 GB, 1.5 x its 2.6 GB of
# algorithmic bytes, 4.5 TB/s) is the physically possible reading.  Uncalibrated either way: the row is a bracket [x1, x2], not evidence.
READ_FACTOR = {'headx3_fused': 1.3, 'head_fused': 1.3, 'conv<fp16x3,k3,s2,NI2,MI4,G2>': 1.0}
f = {label(k): v for k, v in f.items()}; w = {label(k): v for k, v in w.items()}
for k in f:
    if k not in w: continue
    fk, wk = f[k]['FETCH_SIZE'], w[k]['WRITE_SIZE']
    out[k] = (READ_FACTOR.get(k, 2.0) * fk + wk) * 1024
    rows.append((out[k], k, fk, wk))
out['read_factor'] = dict(READ_FACTOR, default=2.0)
json.dump(out, open(os.path.join(root, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
with open(os.path.join(root, 'profiles', f'{tag}_pmc_hbm_traffic.md'), 'w') as md:
    md.write(f'# HBM traffic per launch from PMC counters ({tag})\n\n'
             f'Two separate passes (`rocprofv3 --pmc FETCH_SIZE --kernel-trace -M --output-format csv -- python tools/dev_bench.py {batch} <fp16x3 | bf16> 1`,\n'
             'same with `WRITE_SIZE`; tools/pmc_pass.sh).  FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of a\n'
             'wide coalesced read stream, so reads are doubled (MI355X_MICROARCH.md) -- except for the fused heads, whose isolated 64-byte box\n'
             'pieces are counted in full (x1.3 for their mix, tools/dev/fetch_probe.hip), and for the stem\'s second convolution\n'
             '(`conv<fp16x3,k3,s2,NI2,MI4,G2>`), where x2 would exceed the HBM\'s 8 TB/s: x1, uncalibrated, a bracket not evidence;\n'
             'WRITE_SIZE is uncalibrated.\n\n'
             '| kernel | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch | HBM MB/launch |\n|---|---|---|---|\n')
    for b, k, fk, wk in sorted(rows, reverse=True)[:24]:
        md.write(f'| `{k}` | {fk:.0f} | {wk:.0f} | {b / 1e6:.1f} |\n')
print(open(os.path.join(root, 'profiles', f'{tag}_pmc_hbm_traffic.md')).read())
