for v in SCGLOBAL O2; do echo "== $v"; for i in 1 2 3; do VRA_LIB=$PWD/scratch/variants/lib_$v.so VRA_FORCE_SPLITK=1 python scratch/dbg_gemm2.py 32 2>&1 | grep -E "ones|random bad" ; done; done
