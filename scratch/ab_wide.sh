#!/bin/bash
for w in "46,46" "54,54" "62,62" "46,38"; do
  echo "== MOCAP_WIDE_QUADS=$w"
  MOCAP_WIDE_QUADS=$w bash scratch/ab_args.sh "--markers 8" "--markers 32" "--dist zero" 2>&1 | head -3
done
