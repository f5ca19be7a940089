# kernel timeline of ONE adapter call (fwd+bwd) at a shape: bash tools/timeline_one.sh N C No Co [tag]
cd /tmp; export TMPDIR=/tmp
N=${1:-144}; C=${2:-512}; No=${3:-256}; Co=${4:-384}; TAG=${5:-$N}
rm -rf /tmp/tl; timeout 120 rocprofv3 --kernel-trace -d /tmp/tl -o p -- python $GRAFT_REPO_ROOT/tools/trace_adapter.py $N $C $No $Co 160 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find /tmp/tl -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/timeline_$TAG.txt
