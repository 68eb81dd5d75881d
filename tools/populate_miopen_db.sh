#!/bin/bash
# Populate MIOpen's USER find-db (sta/data/miopen_userdb) with the solver measurements of the configurations the bench,
# the entry points and the tests run, on one MI355X box; copy the result to gpurun_out/miopen_userdb for committing.
# MIOpen appends to the db named after its own build id; runs are cumulative. Usage (GPU box, repo root): bash tools/populate_miopen_db.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export MIOPEN_USER_DB_PATH=$R/diffusion-spacetime-attn_amd/sta/data/miopen_userdb
run() { echo "== $*"; ( time timeout 1500 python bench.py --no-cpu-baseline --no-side-runs --no-roofline --steps 1 "$@" > /dev/null 2> /tmp/pop.log ) 2>&1 | grep real; grep "warm-up" /tmp/pop.log; }
run                                        # fp16, 32 prompts per step (the default bench)
run --dtype bf16
run --images-per-step 16
run --images-per-step 16 --dtype bf16
run --images-per-step 8
run --images-per-step 1
run --images-per-step 24
run --opt-epochs 3 --images-per-step 2     # tracked epochs: backward convolutions, NCHW trunk
run --opt-epochs 3 --images-per-step 1
run --res 768 --objects 4 --images-per-step 4
mkdir -p gpurun_out/miopen_userdb && cp $MIOPEN_USER_DB_PATH/*.txt gpurun_out/miopen_userdb/ && wc -l gpurun_out/miopen_userdb/*.txt
