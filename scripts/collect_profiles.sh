#!/bin/bash
# Collect the rocprofv3 evidence of the bench command on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats        -> per-kernel durations (headline + realistic mix + aggregate kernels)
#   2. --pmc FETCH_SIZE              -> HBM/fabric read bytes   (separate pass, kernel-trace only)
#   3. --pmc WRITE_SIZE              -> HBM/fabric write bytes  (separate pass, kernel-trace only)
#   4. --pmc MFMA busy counters
#   5. --kernel-trace --stats of scripts/aux_kernels.py (noyaux, glitter, land, Mie, profile, absprofile, trphi kernels)
# Outputs land under gpurun_out/prof/<tag>/; scripts/summarize_profiles.py turns them into profiles/<tag>_*.
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
# two legs, profiled separately so that every kernel name has ONE batch size per pass:
#   headline: bench.py --no-mix (k_sos_os, default 32768 bins per step)     realistic: bench.py --workload realistic (k_sos_stream, 4096 bins)
ARGS="$REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-mix"
ARGR="$REPO/bench.py --steps 3 --warmup 1 --no-cpu --workload realistic --bins 4096"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ARGS > $OUT/stats.log 2>&1 && echo stats done
rocprofv3 --kernel-trace --stats -d $OUT/stats_r -o stats_r -- python3 $ARGR > $OUT/stats_r.log 2>&1 && echo stats_r done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1 && echo fetch done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_r -o fetch_r -- python3 $ARGR > $OUT/fetch_r.log 2>&1 && echo fetch_r done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $ARGS > $OUT/write.log 2>&1 && echo write done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_r -o write_r -- python3 $ARGR > $OUT/write_r.log 2>&1 && echo write_r done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o mfma -- python3 $ARGS > $OUT/mfma.log 2>&1 && echo mfma done
rocprofv3 --kernel-trace --stats -d $OUT/aux -o aux -- python3 $REPO/scripts/aux_kernels.py > $OUT/aux.log 2>&1 && echo aux done
#   6. --kernel-trace --stats of the hyperspectral leg (bench.py --workload hyperspectral: sos_spectrum over 2496 wavelengths)
rocprofv3 --kernel-trace --stats -d $OUT/hyper -o hyper -- python3 $REPO/bench.py --workload hyperspectral > $OUT/hyper.log 2>&1 && echo hyper done
find $OUT -name "*.db" | head -20
