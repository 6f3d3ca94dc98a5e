# dev: the long fuzz runs + the stage profiles (second GPU call of the round's evidence); exits non-zero on a GPU fault
O=gpurun_out
timeout -k 10 300 python tools/fuzz_knn.py 300 12 > $O/r06_knn_fuzz.txt 2>&1; rc=$?; tail -2 $O/r06_knn_fuzz.txt
if [ $rc != 0 ] || grep -q "core dump\|Memory access fault" $O/r06_knn_fuzz.txt; then echo "kNN fuzz FAILED"; exit 1; fi
bash tools/dev_final_b.sh > $O/r06_final_b.log 2>&1; rc=$?; echo "b rc=$rc"; tail -30 $O/r06_final_b.log
if grep -q "core dump\|Memory access fault" $O/r06_final_b.log; then exit 1; fi
exit $rc
