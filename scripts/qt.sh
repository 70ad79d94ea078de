for t in ${QT_LIST:-0}; do
  echo "QTICKS=$t"
  MVHMR_QTICKS=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-check --steps 20 --warmup 5 ${QT_ARGS:-} > gpurun_out/qt_$t.json && python scripts/show_bench.py gpurun_out/qt_$t.json
done
