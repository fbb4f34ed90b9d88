cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3f; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "ndt or device_clouds or abi or edges or schemes" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
