#!/bin/bash
# tools/launch_mgpu.sh N [args...] -- the 1/2/4/8-GPU runs of BASELINE config 5 on one node, one process per GPU.
#   default: the driver's bench (python, torch.distributed over RCCL):   tools/launch_mgpu.sh 8
#   C++ host + RCCL through the C-ABI (no Python):                         PIRIP_MGPU=cpp tools/launch_mgpu.sh 8 --streams 6144
N=${1:-1}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "${PIRIP_MGPU:-py}" = "cpp" ]; then
  ID=/tmp/pirip_rccl_id.$$; rm -f $ID
  export PIRIP_RCCL_SESSION=launch$$     # ranks only accept a unique-id file that carries this run's tag
  pids=()
  for r in $(seq 0 $((N-1))); do RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r $R/pirip_amd/bin/mgpu_receiver --id-file $ID "$@" & pids+=($!); done
  rc=0; for p in "${pids[@]}"; do wait $p || rc=$?; done; exit $rc
elif [ "$N" = "1" ]; then
  exec python $R/bench.py --gpus 1 "$@"
else
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29533} $R/bench.py --gpus $N "$@"
fi
