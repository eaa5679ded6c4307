#!/bin/bash
# A/B of the single-strip X sweep (no prefetch buffer, one strip per wave): tuned, tuned + dt tracking, exact, exact + tracking
V=variants
echo "== tuned, no tracking"; python tools/ab_sweep.py --rounds 15 --copy base=$V/xs0/libarmon_hip.so single=$V/xs7/libarmon_hip.so | grep sweep_X
echo "== tuned, dt tracking on X"; python tools/ab_sweep.py --rounds 15 --track-x base=$V/xs0/libarmon_hip.so single=$V/xs3/libarmon_hip.so | grep sweep_X
echo "== exact, no tracking"; python tools/ab_sweep.py --rounds 15 --exact base=$V/xs0/libarmon_hip.so single=$V/xs3/libarmon_hip.so | grep sweep_X
echo "== exact, dt tracking on X"; python tools/ab_sweep.py --rounds 15 --exact --track-x base=$V/xs0/libarmon_hip.so single=$V/xs3/libarmon_hip.so | grep sweep_X
echo "== tuned, 4096x8192 tile"; python tools/ab_sweep.py --rounds 30 --shape 4096x8192 --copy base=$V/xs0/libarmon_hip.so single=$V/xs7/libarmon_hip.so | grep sweep_X
