for f in "-DLK_DBG_SELF" "-DLK_DBG_ATOMIC" ""; do
  echo "=== flags: $f"
  rm -f lkpy_amd/csrc/_obj/iknn_build.o
  LK_EXTRA_FLAGS="$f" python __graft_entry__.py 2>&1 | grep -E "error|built" | cut -c1-60
  timeout 100 python -m pytest tests/test_gpu_iknn.py -x -q -k "ml_small" 2>&1 | tail -1
done
