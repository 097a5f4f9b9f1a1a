#!/bin/bash
# The rocprofv3 passes behind roofline.traffic, for one configuration (run from the repository root on a GPU box):
#     tools/run_pmc.sh <C2|C3|C4bf|C5|C5_unstructured|EXP_C1|EXP_1000_tips> [replicas | samples]
# one --kernel-trace --stats pass and one --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; the guide's
# rule: counters in runs of their own, never together with a trace), the program itself after `--`.  Everything lands under
# gpurun_out/pmc_<name>/ and gpurun_out/r04_*: copy the summaries into profiles/ and fold them with `python tools/pmc_summary.py --merge profiles`.
set -e
name=$1; shift
R=$(pwd)
out=$R/gpurun_out/pmc_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
T="python3 $R/tools/pmc_target.py $name $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $T > $out/target.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- $T > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- $T > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $out/pmc_sq -o p -- $T > /dev/null
cd $R
units=$(python3 -c "import json,sys; print(json.loads(open('$out/target.json').read().strip().splitlines()[-1])['units_per_sweep'])")
python3 tools/pmc_summary.py $name $units 14 3 $out $R/gpurun_out | tail -12
cp $(ls $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/r04_kernel_stats_$name.csv 2>/dev/null || true
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/trace      # raw passes: tens of MB; the summaries are what is kept
