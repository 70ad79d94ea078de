for t in ${QT_LIST:-0}; do
  echo "DBG=$t"
  MVHMR_QTICKS=$t timeout -k 10 200 python scripts/bench_bwd.py --iters 2 2>&1 | grep fwd | tail -1
done
