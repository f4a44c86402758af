for f in tests/test_focf_hip.py tests/test_trainer_hip.py tests/test_primitives_hip.py tests/test_nfcf_hip.py tests/test_sharded_hip.py; do
  echo "== $f"; timeout 200 python -m pytest $f -x -q --timeout 60 2>&1 | grep -E "passed|failed|rror|Timeout" | tail -4
done
