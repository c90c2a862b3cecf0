#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
for nth in 0 1 7; do
SNCAL_TT_TRACE=$O/tt_c32_$nth.bin SNCAL_TT_TRACE_NTH=$nth python tools/dev/tt_trace_run.py bf16x3 64 > $O/tt.log 2>&1
(echo "=== c32 launch $nth"; python tools/tt_trace.py $O/tt_c32_$nth.bin; python tools/tt_pipe.py $O/tt_c32_$nth.bin | head -8) > $O/tt_c32_$nth.txt 2>&1
SNCAL_TT_TRACE=$O/tt_c23_$nth.bin SNCAL_TT_TRACE_CFG64=1 SNCAL_TT_TRACE_NTH=$nth python tools/dev/tt_trace_run.py bf16x3 64 > $O/tt.log 2>&1
(echo "=== c23 launch $nth"; python tools/tt_trace.py $O/tt_c23_$nth.bin; python tools/tt_pipe.py $O/tt_c23_$nth.bin | head -8) > $O/tt_c23_$nth.txt 2>&1
done
cat $O/tt_c32_*.txt $O/tt_c23_*.txt
rm -f $O/*.bin
