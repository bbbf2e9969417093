cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=gpurun_out/pmc_w2b; mkdir -p $R/$out
rocprofv3 -L > $R/$out/counters.txt 2>&1
grep -o "SQ_LDS[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_INST_CYCLES_VMEM\|SQ_INSTS_SALU\|SQ_VALU_MFMA[A-Z_]*\|SQ_INST_LEVEL_LDS" $R/$out/counters.txt | sort -u > $R/$out/lds_counters.txt
cat $R/$out/lds_counters.txt
BFM_W2_DBG=23 rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/scripts/run_one_conv.py 3 160 64 64 3 > $R/$out/p1.log 2>&1
BFM_W2_DBG=23 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/$out/p2 -o p2 --output-format csv -- python3 $R/scripts/run_one_conv.py 3 160 64 64 3 > $R/$out/p2.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
tot = collections.OrderedDict(); n = {}
for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv_wino2" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k, v in tot.items(): print("%-28s %18.0f per dispatch" % (k, v / n[k]))
PY
tail -2 $out/p1.log
