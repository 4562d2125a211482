cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r4k
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r4k/sq_v3 -- python $R/tools/satd_only.py 3 20 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/r4k/lds_v3 -- python $R/tools/satd_only.py 3 20 > /dev/null 2>&1
echo done
