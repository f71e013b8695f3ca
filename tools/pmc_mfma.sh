#!/bin/bash
# MFMA-pipe utilisation and LDS bank conflicts of conv_gemm_f32 (one counter set per pass, kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "MFMA|BANK_CONFLICT|SQ_BUSY_CYCLES|LDS_IDX|GRBM_GUI_ACTIVE" | head -40 > $O/pmc_avail.txt
rm -rf $O/pmc_mfma $O/pmc_lds
TS_TUNE_FEW=1 TS_B=64 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python $R/tools/tune_conv.py > $O/pmc_mfma.log 2>&1
TS_TUNE_FEW=1 TS_B=64 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds -- python $R/tools/tune_conv.py > $O/pmc_lds.log 2>&1
find $O/pmc_mfma $O/pmc_lds -name "*kernel_trace.csv" -delete
tail -2 $O/pmc_mfma.log | cut -c1-200; wc -l $O/pmc_avail.txt
