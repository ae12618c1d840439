cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_PORT=29833 WORLD_SIZE=2
for r in 0 1; do RANK=$r python tools/dbg_dist2.py 2>/dev/null > /tmp/dd_$r.log & done; wait
cat /tmp/dd_0.log /tmp/dd_1.log | grep rank | cut -c1-900
