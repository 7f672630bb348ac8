set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/final3
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02/final3/smoke.log 2>&1 || (tail -20 gpurun_out/r02/final3/smoke.log; exit 1)
tail -3 gpurun_out/r02/final3/smoke.log
