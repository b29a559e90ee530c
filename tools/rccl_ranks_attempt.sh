#!/bin/bash
# tools/rccl_ranks_attempt.sh -- what a 1-GPU gpurun lease allows towards an RCCL run with more than one rank.
# Round 6's bounded attempt: an MI355X can be split into 8 logical devices (compute partition mode CPX, one XCD each), over
# which bench.py --gpus 8 and the C-only *_multi entry points would meet RCCL with D > 1.  The pool REFUSES any call whose
# script switches the partition mode (the refusal text, verbatim, heads profiles/r06_rccl_ranks.txt), so this script only READS
# the mode and records what RCCL says to two ranks on the one device.  Transcript: gpurun_out/r06/rccl_ranks.txt.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${ROUND:-r06}
mkdir -p $OUT
LOG=$OUT/rccl_ranks.txt
cd $REPO
exec > >(tee $LOG) 2>&1
SMI="timeout 60 rocm-smi"
echo "# tools/rccl_ranks_attempt.sh on $(hostname), $(date -u +%FT%TZ), uid $(id -u)"
echo "## compute / memory partition mode of the lease (read only)"
$SMI --showcomputepartition; echo "rc=$?"
$SMI --showmemorypartition; echo "rc=$?"
ls -l /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition 2>&1
cat /sys/class/drm/card*/device/available_compute_partition 2>&1
python - <<'EOF2'
import torch
print("torch sees", torch.cuda.device_count(), "device(s):",
      [(torch.cuda.get_device_properties(i).name, torch.cuda.get_device_properties(i).multi_processor_count,
        str(torch.cuda.get_device_properties(i).uuid)) for i in range(torch.cuda.device_count())])
print("RCCL", torch.cuda.nccl.version())
EOF2
echo "## two ranks, ONE device, backend nccl: what RCCL answers (why the self-launch tests gather over gloo)"
timeout 150 python - <<'EOF2' 2>&1 | tail -15
import os, subprocess, sys
code = r'''
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=2, device_id=torch.device("cuda", 0))
try:
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", os.environ["RANK"], "all_reduce ok?!", t.tolist())
except Exception as e:
    print("rank", os.environ["RANK"], "RCCL refused:", " | ".join(str(e).splitlines())[:600])
'''
env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
ps = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
for p in ps:
    try:
        out, _ = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        p.kill(); out = "(timed out)"
    keep = [l for l in out.splitlines() if "RCCL refused" in l or "ok?!" in l or "Duplicate" in l or "invalid usage" in l.lower()]
    print("\n".join(keep)[:1500] if keep else out[-800:])
EOF2
echo "## RESULT: no RCCL run with more than one rank is possible on this lease (1 device, partition mode fixed by the pool)."
