# PMC evidence for one kernel of one command: two separate passes, --kernel-trace only (rocprofv3 refuses --pmc with
# the trace domains on this pool).  usage: bash scripts/pmc_kernel.sh <outdir> <kernel name substrings, comma separated> <python script + args>
set -e
out=$1; shift
kern=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
script=$1; shift
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_VALU -d $R/$out/p1 -o p1 --output-format csv -- python3 $R/$script "$@" > $R/$out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $R/$out/p2 -o p2 --output-format csv -- python3 $R/$script "$@" > $R/$out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d $R/$out/p3 -o p3 --output-format csv -- python3 $R/$script "$@" > $R/$out/p3.log 2>&1 || true
cd $R
python3 - <<PY
import csv, glob, collections
kerns = "$kern".split(",")
with open("$out/summary.txt", "w") as fo:
    for kern in kerns:
        tot = collections.OrderedDict()
        n = {}
        for f in sorted(glob.glob("$out/p*/**/*counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                if kern not in r["Kernel_Name"]:
                    continue
                k = r["Counter_Name"]
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
                n[k] = n.get(k, 0) + 1
        fo.write("== %s\n" % kern)
        for k, v in tot.items():
            fo.write("%-28s %18.0f   (%d dispatches, %.0f per dispatch)\n" % (k, v, n[k], v / n[k]))
print(open("$out/summary.txt").read())
PY
tail -3 $out/p1.log
