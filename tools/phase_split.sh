#!/bin/bash
# Phase split (wave-cycles) and wave-level block counts of the segment loop: needs variants/phase.so = libmirt.so built with -DMIRT_PHASE_TIMING.
# usage: tools/phase_split.sh <outfile> [bench args...]
out=$1; shift
# variants/counts.so = the same with -DMIRT_BLOCK_COUNTS (its atomics distort the cycle split, so the two are separate builds)
: > $out
for v in ${PHASE_VARIANTS:-phase counts}; do  # PHASE_VARIANTS="phase_base counts_base": the same builds of another source state
  [ -f variants/$v.so ] || continue
  MIRT_LIB=$PWD/variants/$v.so timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $out.err || { tail -5 $out.err; exit 1; }
  if [ "${v#phase}" != "$v" ]; then grep "^\[mirt phase" $out.err | tail -5 >> $out; else grep "^\[mirt block" $out.err | awk '!seen[$0]++' | tail -24 >> $out; fi
done
cat $out
