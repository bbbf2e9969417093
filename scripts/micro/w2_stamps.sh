# Diagnostic build of the library with -DBFM_W2_STAMPS: workgroup 0 of conv_wino2 records the shader clock around every
# barrier (multiplying wave 0, staging wave 4); prints per-tap phases.  usage (on the GPU box): bash scripts/micro/w2_stamps.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/w2s
objs=""
for f in brainfm_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = conv3d_wino ]; then
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -DBFM_W2_STAMPS -c $f -o gpurun_out/w2s/$b.o
    objs="$objs gpurun_out/w2s/$b.o"
  else
    objs="$objs brainfm_amd/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o gpurun_out/w2s/libstamps.so
BFM_LIB_PATH=$R/gpurun_out/w2s/libstamps.so BFM_W2_STAMP_FILE=$R/gpurun_out/w2s/stamps.txt BFM_W2_DBG=${1:-0} timeout -k 10 120 python3 scripts/run_one_conv.py 3 160 64 64 2
python3 - <<PY
import collections
rows = collections.defaultdict(dict)
for l in open("$R/gpurun_out/w2s/stamps.txt"):
    r, k, t = l.split()
    rows[int(r)][int(k)] = int(t)
for r in (0, 1):
    ts = [rows[r][k] for k in sorted(rows[r]) if rows[r][k]]
    t0 = ts[0]
    # pairs (arrive, release): k even = arrive, odd = release
    out = []
    for i in range(0, min(len(ts) - 2, 2 * 46), 2):
        arrive, release, nxt = ts[i], ts[i + 1], ts[i + 2]
        out.append("%5d w%4d r%5d" % (arrive - t0, release - arrive, nxt - release))
    print("role", r, "(arrive-t0, wait at barrier, run until next barrier):")
    for i in range(0, len(out), 4):
        print("   ", " | ".join(out[i:i + 4]))
PY
