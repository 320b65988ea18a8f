for r in 1 2 3 4; do
  for m in 0 1; do
    if [ $m == 1 ]; then export FR_NO_SPECULATION=1; else unset FR_NO_SPECULATION; fi
    echo "nospec=$m $(python tools/frames.py 180 0 fov 2>/dev/null | tail -1) | $(python tools/frames.py 90 0 pcheck_obb 2>/dev/null | tail -1)"
  done
done
