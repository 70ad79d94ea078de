for t in ${QT_LIST:-0 200 218 240 300 436}; do
  echo "QTICKS=$t"
  MVHMR_QTICKS=$t timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/qt_$t.json && python scripts/show_bench.py gpurun_out/qt_$t.json
done
