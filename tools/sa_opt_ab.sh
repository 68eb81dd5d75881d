# A/B of the level-0 self-attention launch: standard loop vs optimistic loop (+ empty repair launch), both 16-bit types, interleaved on one box
cd $GRAFT_REPO_ROOT
for r in 1; do
for dt in fp16 bf16; do for o in 0 1; do for qs in 0.25 1.0; do
STA_SA_QSCALE=$qs STA_SA_DTYPE=$dt STA_SA_OPT=$o python tools/selfattn_l0_time.py 2>&1 | tail -1
done; done; done; done
