#!/bin/bash
# final trace line of every image under different engine knobs
run() { echo "== $*"; env "$@" python tools/cli_survey_bench.py 1920 1080 6 4 24 2>&1 | grep -E "iter: 0199|iter: 01[0-9][0-9], cost: (inf|nan)" | cut -c1-110 | head -30; }
run SUCRE_IMAGES_IN_FLIGHT=2 SUCRE_CULL_VIEWS=1
run SUCRE_IMAGES_IN_FLIGHT=1 SUCRE_CULL_VIEWS=1
run SUCRE_IMAGES_IN_FLIGHT=1 SUCRE_CULL_VIEWS=0
