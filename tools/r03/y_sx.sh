#!/bin/bash
# the Y march's LDS store exchange: automatic / forced / off across row pitches, fp64 and fp32 (tools/r03/f32_pitch.py)
T=tools/r03/f32_pitch.py
echo "== fp64 automatic";           python $T --dtype float64 --cases 16384:4,16388:4,16387:4,16386:4
echo "== fp64 never (ARMON_Y_SX=2)"; ARMON_Y_SX=2 python $T --dtype float64 --cases 16388:4,16387:4
echo "== fp64 always (ARMON_Y_SX=1)"; ARMON_Y_SX=1 python $T --dtype float64 --cases 16384:4
echo "== fp32 automatic";           python $T --cases 16384:4,16392:4,16387:4,16388:4
echo "== fp32 never";               ARMON_Y_SX=2 python $T --cases 16384:4,16387:4
echo "== fp32 always";              ARMON_Y_SX=1 python $T --cases 16392:4
