for v in "" few8 few16; do
  if [ -n "$v" ]; then export MIRT_LIB=$PWD/variants/$v.so; else unset MIRT_LIB; fi
  echo "== ${v:-product}"
  timeout -k 10 200 python3 tools/cull_probe.py 2>&1 | grep "f32 tile cull" || exit 1
  timeout -k 10 200 python3 tools/inflight_probe.py 8 60 2>&1 | grep "part 0" || exit 1
done
