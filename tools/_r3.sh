cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "device_clouds or schemes" > $O/t.log 2>&1 || { tail -80 $O/t.log; exit 1; }
tail -3 $O/t.log
RSREG_INC_VERBOSE=1 python tools/cpp_scheme_times.py N300 16 2>&1 | grep "device clouds" > $O/inc.txt
grep -c merged $O/inc.txt; grep "run" $O/inc.txt
RSREG_NO_INCREMENTAL=1 python tools/cpp_scheme_times.py N300 16 2>&1 | grep "device clouds" | grep run > $O/noinc.txt; cat $O/noinc.txt
