set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/final3
export TMPDIR=/tmp
O=gpurun_out/r02/final3
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/gputests_abi3.log 2>&1 || (tail -30 $O/gputests_abi3.log; exit 1)
tail -2 $O/gputests_abi3.log
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/lp_fetch --output-format csv -- python3 tools/lowpass_pmc_run.py > $O/lp_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/lp_write --output-format csv -- python3 tools/lowpass_pmc_run.py > $O/lp_write.log 2>&1
python tools/pmc_summary.py $O/pmc_praat_lowpass.json lp_ $O/lp_fetch $O/lp_write
rm -rf $O/lp_fetch $O/lp_write
cat $O/pmc_praat_lowpass.json | head -60
