# Section-slot read path with parts of the section kernel switched off (probe build): 0 = whole kernel, 1 = stage the
# bytes but skip the octet loop, 5 = the octet loop without HBM traffic (spans of zero bytes).
set -e
export CVEC_LIB_PATH=$GRAFT_REPO_ROOT/compressed-vec_amd/lib/libcvec_hip_probes.so PROBE_PATH=1
cd $GRAFT_REPO_ROOT
for d in sinusoid splitmix; do
  for a in 0 1 5; do
    CVEC_ABLATE=$a timeout -k 10 120 python tools/vstream_probe.py $d 2>&1 | grep ablate
  done
done
