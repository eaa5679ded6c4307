V=variants
echo "== Sod"; python3 tools/ab_sweep.py --rounds 15 --copy base=$V/r4base/libarmon_hip.so "$@" | grep sweep_Y
