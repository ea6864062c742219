set -e
export CVEC_LIB_PATH=$GRAFT_REPO_ROOT/compressed-vec_amd/lib/libcvec_hip_probes.so
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/vstream_probe.py sinusoid 2>&1 | grep ablate
CVEC_ABLATE=1 timeout -k 10 120 python tools/vstream_probe.py sinusoid 2>&1 | grep ablate
timeout -k 10 120 python tools/vstream_probe.py splitmix 2>&1 | grep ablate
CVEC_ABLATE=1 timeout -k 10 120 python tools/vstream_probe.py splitmix 2>&1 | grep ablate
