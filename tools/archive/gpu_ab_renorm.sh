for cfg in "--dtype f64" "--dtype f64 --mode spot" "--mode last" "--mode spot" ""; do echo "#### $cfg"; EXTRA="$cfg" bash tools/gpu_variant.sh norenorm 2>&1 | grep -v "^$"; done
