#!/bin/bash
tag=r04v21
PMC_B=64 PMC_DTYPE=fp16x3 bash tools/pmc_pass.sh $tag/pmc "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" > gpurun_out/$tag.pmc.log 2>&1
tail -4 gpurun_out/$tag.pmc.log
