#!/bin/bash
# prof_cmd.sh <tag> <python-script> [args...] -- rocprofv3 of one tool script: kernel stats, then separate PMC passes
# (FETCH_SIZE | WRITE_SIZE | SQ busy counters), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Output: gpurun_out/<tag>_*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 "$@" > gpurun_out/${tag}_stats.log 2>&1
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
find gpurun_out/${tag}_stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 "$@" > /dev/null 2> gpurun_out/${tag}_fetch.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 "$@" > /dev/null 2> gpurun_out/${tag}_write.err
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_sq -- python3 "$@" > /dev/null 2> gpurun_out/${tag}_sq.err
python3 green-marl_amd/tools/pmc_report.py ${tag} > gpurun_out/${tag}_pmc.txt 2>&1
for d in fetch write sq; do find gpurun_out/${tag}_$d -name "*.csv" -size +8M -delete; done
cat gpurun_out/${tag}_pmc.txt
