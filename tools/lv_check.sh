#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lv1; mkdir -p $O; cd $R
./build/ubench_valu > $O/ubench.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_fullsize.py tests/test_gpu_small.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
tail -5 $O/pytest.log
bash tools/quick_ab.sh tree 2>&1 | tee -a $O/summary.txt
echo "planes off:" | tee -a $O/summary.txt; ANX_FS_PLANES=0 bash tools/quick_ab.sh tree 2>&1 | tail -1 | tee -a $O/summary.txt
