#!/bin/bash
# Round-end verification on the GPU box: see tools/r03_verify.sh (GPU tests, smoke, bench with PMC + CPU leg, bench under the kernel tracer,
# the other BASELINE configurations, the RCCL path with one rank); tools/r03_soak.sh adds a 200-step run and the torchrun launcher.
bash "$(dirname "$0")/r03_verify.sh"
