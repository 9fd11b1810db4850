#!/bin/bash
# The round-3 profile set: kernel stats, SQ counters, HBM bytes and the phase split for every workload bench.py reports a roofline for.
set -o pipefail
only=" $* "   # optional: the tags to redo (default all)
run() { tag=$1; shift; [ "$only" = "  " ] || [[ "$only" == *" $tag "* ]] || return 0; tools/profile.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1 || { echo "profile $tag failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }; tools/phase_split.sh gpurun_out/prof_$tag/phase_split.txt "$@" > /dev/null || exit 1; echo "done $tag"; }
run mx
run valu --accel valu
run stress10k --workload stress10000 --width 3840 --spp 16
run stress10k_valu --workload stress10000 --width 3840 --spp 16 --accel valu
run textured --workload textured --width 3840 --spp 256
