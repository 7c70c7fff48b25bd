# usage: bash tests/probes/pmc_cmd.sh <tag> <kernel name pattern> <command...>   -- SQ counter passes of an arbitrary command (GPU box)
TAG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b $R/gpurun_out/pmc_c
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_a -o a -- "$@" > $R/gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_b -o b -- "$@" > $R/gpurun_out/pmc_b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_c -o c -- "$@" > $R/gpurun_out/pmc_c.log 2>&1
cd $R; python pathfinder.jl_amd/tools/pmc_sq.py "$TAG" "gpurun_out/pmc_a/*.db" "gpurun_out/pmc_b/*.db" "gpurun_out/pmc_c/*.db" > gpurun_out/pmc_$TAG.md; grep -i "$PAT" gpurun_out/pmc_$TAG.md
