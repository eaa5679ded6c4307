#!/bin/bash
# Round-2 A/B matrix of the Y march (run on the GPU box from the repo root): builds in variants/ made by
# tools/build_variant.sh <name> -DARMON_ONLY_HEADLINE <flags>; run lengths through the ARMON_Y_SEG knob.
set -e
V=variants
L=libarmon_hip.so
python tools/ab_sweep.py --rounds 15 --copy \
  base=armon.jl_amd/$L hl=$V/hl/$L noc=$V/noc/$L pf5=$V/pf5/$L pf3=$V/pf3/$L pf2=$V/pf2/$L w3=$V/w3/$L w3pf3=$V/w3pf3/$L \
  b128=$V/b128/$L b512=$V/b512/$L \
  s64=$V/hl/$L s256=$V/hl/$L s421=$V/hl/$L s529=$V/hl/$L s713=$V/hl/$L s1093=$V/hl/$L s2341=$V/hl/$L \
  n529=$V/noc/$L n2341=$V/noc/$L \
  --env "s64:ARMON_Y_SEG=64;s256:ARMON_Y_SEG=256;s421:ARMON_Y_SEG=421;s529:ARMON_Y_SEG=529;s713:ARMON_Y_SEG=713;s1093:ARMON_Y_SEG=1093;s2341:ARMON_Y_SEG=2341;n529:ARMON_Y_SEG=529;n2341:ARMON_Y_SEG=2341"
