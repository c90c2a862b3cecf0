#!/bin/bash
# round 5: the PMC passes of the final build (traffic + matrix-pipe utilisation), then the bench lines and the kernel stats
export PMC_B=64 PMC_DTYPE=fp16x3
bash tools/pmc_pass.sh r05v26_x3 FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" 2>&1 | tail -12
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py gpurun_out/r05v26_x3 r05v26 64 > gpurun_out/r05v26_x3/traffic.md 2>&1; tail -24 gpurun_out/r05v26_x3/traffic.md
python tools/pmc_mfma.py gpurun_out/r05v26_x3 r05v26_fp16x3 2>&1 | tail -14
cp profiles/pmc_traffic.json profiles/r05v26_pmc_hbm_traffic.md profiles/r05v26_fp16x3_pmc_mfma_util.* gpurun_out/r05v26_x3/ 2>/dev/null
