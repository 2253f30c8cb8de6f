#!/bin/bash
# A/B of the moment-family evaluation (quads / pairs) and of the bucket-direct sampler's forms, warm device.
set -e
echo "== default"; python tools/ab_cdf_warm.py
echo "== philox"; MCX_AB_RNG=philox python tools/ab_cdf_warm.py
echo "== philox, guided search"; MCX_NO_DIRECT=1 MCX_AB_ONLY=beta MCX_AB_RNG=philox python tools/ab_cdf_warm.py
echo "== philox, no register cap"; MCX_AB_ONLY=beta MCX_AB_RNG=philox MCX_EXTRA_DEFINES="MCX_INTEGRATE_ATTR=" python tools/ab_cdf_warm.py
