V=variants
echo "== Sod"; python3 tools/ab_sweep.py --rounds 15 --copy base=$V/r4base/libarmon_hip.so pre=$V/r4pre/libarmon_hip.so prio=$V/r4prio/libarmon_hip.so | grep sweep_X
echo "== Biz"; python3 tools/ab_sweep.py --rounds 15 --test Bizarrium base=$V/r4base_biz/libarmon_hip.so pre=$V/r4pre_biz/libarmon_hip.so prio3=$V/r4prio_biz/libarmon_hip.so prio1=$V/r4prio1_biz/libarmon_hip.so | grep sweep_X
