#!/bin/bash
# the window width at the thresholds of the rule (c = 15 / 16 at 2^19, 16 / 17 at 2^21), re-measured on the round-5 binary: blocking and in flight
for lg in 18 19 20 21 22; do python tools/dbg/window_blocking.py $lg 15 16 17 18 2>&1 | grep -v amdgpu.ids; done
for lg in 18 19 20 21 22; do python tools/dbg/window_pipe.py $lg 15 16 17 18 2>&1 | grep -v amdgpu.ids | tail -4; done
