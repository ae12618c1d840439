cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=0 WORLD_SIZE=1 MASTER_PORT=29811 python tests/dist_sampler_worker.py /tmp/one_new 2>/dev/null
BPA_SMP_V1=1 RANK=0 WORLD_SIZE=1 MASTER_PORT=29812 python tests/dist_sampler_worker.py /tmp/one_old 2>/dev/null
for r in 0 1; do RANK=$r WORLD_SIZE=2 MASTER_PORT=29813 python tests/dist_sampler_worker.py /tmp/two_new 2>/dev/null & done; wait
for r in 0 1; do BPA_SMP_V1=1 RANK=$r WORLD_SIZE=2 MASTER_PORT=29814 python tests/dist_sampler_worker.py /tmp/two_old 2>/dev/null & done; wait
python - <<PY
import json
a=json.load(open("/tmp/one_new.0.json")); b=json.load(open("/tmp/one_old.0.json"))
c=json.load(open("/tmp/two_new.0.json")); d=json.load(open("/tmp/two_old.0.json"))
for nm,x in (("one_new",a),("one_old",b),("two_new",c),("two_old",d)):
    print(nm, x["taus"][4:], x["thetas"][4:], x["summary"])
PY
