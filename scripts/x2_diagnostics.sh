#!/bin/bash
# the known-answer validation (scripts/validate_posterior.py, n_pix 1024, batch 8 as bbhMahoGANy.py:84-89) three times on ONE trained CNN:
# product defaults; BatchNormalization moving statistics as plain EMA; DIAGNOSTIC binary cross-entropy on the discriminator logit.  One GPU call (~15 min).
set -e
out=${1:-gpurun_out/x2}
mkdir -p $out
python scripts/validate_posterior.py --pe-batch 8 --pe-iter 500000 --cnn-seconds ${CNN_SECONDS:-420} --graph --gan-batch 8 --gan-iter ${GAN_ITER:-20000} --gan-seconds 170 \
    --cadence 1000 --save-pe $out/pe.h5 --out $out/default.json > $out/default.log 2>&1
python scripts/validate_posterior.py --load-pe $out/pe.h5 --graph --gan-batch 8 --gan-iter ${GAN_ITER:-20000} --gan-seconds 170 --cadence 1000 --moving-average ema \
    --out $out/ema.json > $out/ema.log 2>&1
python scripts/validate_posterior.py --load-pe $out/pe.h5 --gan-batch 8 --gan-iter ${GAN_ITER:-20000} --gan-seconds 170 --cadence 1000 --bce-on-logit \
    --out $out/bce_on_logit.json > $out/bce_on_logit.log 2>&1
rm -f $out/pe.h5
