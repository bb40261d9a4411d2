#!/bin/bash
# tools/gpu_pmc_sq_configs.sh <tag> — the two SQ counter groups for configs[2] (--mapping) and configs[3] (--sensor ROWS128), all aloam kernels
# (tools/gpu_evidence.sh takes them for the headline only); summaries land in gpurun_out/<tag>/pmc_sq{1,2}_{mapping,rows128}.md
TAG=${1:-sqcfg}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "mapping:--mapping" "rows128:--sensor ROWS128"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for grp in "sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    g=${grp%%:*}; ctr=${grp#*:}
    rm -rf /tmp/pmc_${name}_$g
    rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex aloam --output-format csv -d /tmp/pmc_${name}_$g -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 $args > $O/pmc_${g}_$name.log 2>&1
    (cd $R && python tools/pmc_summary.py /tmp/pmc_${name}_$g $O/pmc_${g}_$name.md > /dev/null 2>>$O/pmc_${g}_$name.log)
  done
done
sha256sum $R/a-loam_amd/lib/libaloam_mi355x.so | cut -c1-64 > $O/lib_sha256.txt
ls $O
