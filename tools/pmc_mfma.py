"""Dev helper: matrix-pipe utilisation per kernel from a rocprofv3 counter pass (tools/pmc_pass.sh OUT "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE") -> profiles/<tag>_pmc_mfma_util.md.

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's SIMDs, the cycles a SIMD's matrix pipe is busy (32 per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md).  GRBM_GUI_ACTIVE is the kernel's duration in shader clocks (rocprofv3 sums the 8 XCDs' instances: / 8).
utilisation = MFMA busy cycles / (duration in clocks x 1024 SIMDs)."""
import json, os, sys
d, tag = sys.argv[1], sys.argv[2]
grp = [g for g in os.listdir(d) if 'MFMA_BUSY' in g][0]
s = json.load(open(os.path.join(d, grp, 'summary.json')))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for k, c in s.items():
    busy, gui = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('GRBM_GUI_ACTIVE', 0.0)
    if not busy or not gui:
        continue
    clocks = gui / 8.0
    rows.append((busy, k, busy / (clocks * 1024.0), clocks, c.get('SQ_BUSY_CYCLES', 0.0), c.get('SQ_WAVE_CYCLES', 0.0)))
rows.sort(reverse=True)
out = {k: {'mfma_util': round(u, 4), 'mfma_busy_cycles_per_launch': b, 'clocks_per_launch': cl} for b, k, u, cl, _, _ in rows}
json.dump(out, open(os.path.join(root, 'profiles', f'{tag}_pmc_mfma_util.json'), 'w'), indent=1)
with open(os.path.join(root, 'profiles', f'{tag}_pmc_mfma_util.md'), 'w') as md:
    md.write(f'# Matrix-pipe utilisation per kernel from PMC counters ({tag})\n\n'
             '`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -M --output-format csv -- python tools/dev_bench.py 64 <engine> 1`\n'
             '(its own pass, tools/pmc_pass.sh).  utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of\n'
             'SIMD-cycles in which the matrix pipe holds an MFMA (includes MFMAs on padding pixels; a profiled pass runs at a lower clock than an\n'
             'unprofiled one, the ratio does not depend on it).\n\n'
             '| kernel | MFMA busy cycles / launch | shader clocks / launch | matrix-pipe utilisation |\n|---|---|---|---|\n')
    for b, k, u, cl, _, _ in rows[:12]:
        md.write(f'| `{k}` | {b:.4g} | {cl:.4g} | {u:.3f} |\n')
print(open(os.path.join(root, 'profiles', f'{tag}_pmc_mfma_util.md')).read())
