# A/B of the headline kernel on ONE box (boxes differ by up to 5 %): put the other build at mpc-protocols_amd/libhbmpc_hip_prev.so,
# then  gpurun -- bash tools/ab_headline.sh
set -e
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python -u bench.py --no-extra --cpu-sample-log2 12 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'compute_shares ms/step %.4f kernel %.4f | recon ms/step %.4f kernel %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['recon']['ms_per_step'], j['roofline_recon']['kernel_ms']))" $1; }
cp mpc-protocols_amd/libhbmpc_hip.so /tmp/new.so
run new; run new
cp mpc-protocols_amd/libhbmpc_hip_prev.so mpc-protocols_amd/libhbmpc_hip.so
run prev; run prev
cp /tmp/new.so mpc-protocols_amd/libhbmpc_hip.so
run new
