#!/bin/sh
# drives cli_args_asan (cli/debwt.c over stubs) through its argument and error paths; any sanitizer report fails the run
set -u
X=./cli_args_asan; D=$(mktemp -d); F=$D/in.fa; printf '>a\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n' > $F
bad=0
expect() { want=$1; shift; "$@" > $D/out.txt 2> $D/err.txt; rc=$?; if grep -q "Sanitizer\|runtime error" $D/err.txt; then cat $D/err.txt; bad=1; fi
           if [ $rc -ne $want ]; then echo "exit $rc (want $want): $*"; cat $D/err.txt; bad=1; fi; }
expect 1 $X
expect 1 $X -o $D/o
expect 0 $X -o $D/o $F
expect 0 $X -o $D/o -k 12 -t 3 -j /nowhere --device 0 --iupac 7 $F
expect 1 $X -o $D/o -k 40 $F
expect 1 $X -o $D/o -t zero $F
expect 1 $X -o /nonexistent_dir/o $F
expect 1 $X -o $D/o $D/missing.fa
expect 1 $X -o $D/o --frobnicate 1 $F
expect 0 $X -o $D/o --gpus 4 $F
expect 0 $X -o $D/o --gpus 3 --devices 0,1,2 --keys exchange $F
expect 0 $X -o $D/o --devices 0,0 --keys rescan $F
expect 1 $X -o $D/o --gpus 2 --devices 0,x $F
expect 1 $X -o $D/o --gpus 2 --devices 0,1,2 $F
expect 1 $X -o $D/o --gpus 1 --devices 0, $F
expect 1 $X -o $D/o --gpus 300 $F
expect 1 $X -o $D/o --gpus 2 --keys sideways $F
expect 0 $X -o $D/o --gpus 2 --exchange rccl $F
expect 0 $X -o $D/o --gpus 2 --exchange peer --keys exchange $F
expect 1 $X -o $D/o --gpus 2 --exchange pigeon $F
expect 1 $X -o $D/o --gpus 2 $D/missing.fa
expect 0 $X -o $D/o --verify $F
expect 0 $X --verify -o $D/o -k 16 $F
expect 0 $X -o $D/o --gpus 2 --verify $F
DEBWT_STUB_VERIFY_FAIL=1 expect 1 $X -o $D/o --verify $F
DEBWT_STUB_VERIFY_FAIL=1 expect 1 $X -o $D/o --gpus 2 --verify $F
mkdir $D/dump; expect 0 $X -o $D/o --dump $D/dump --verify $F
[ -f $D/dump/stage1 ] && [ -f $D/dump/stage2 ] && [ -f $D/dump/stage3 ] || { echo "--dump wrote no files"; bad=1; }
expect 1 $X -o $D/o --dump $D/nodir $F
expect 1 $X -o $D/o --gpus 2 --dump $D/dump $F
L=$(python3 -c "print(','.join(['0']*400))"); expect 1 $X -o $D/o --devices $L $F
expect 0 $X -o $D/o --gpus 2 $F          # (a failed run has removed OUT: the reference's create+remove probe, src/main.c:55-58)
[ "$(wc -c < $D/o)" = 32 ] && [ "$(wc -c < $D/o.#)" = 16 ] && [ "$(wc -c < $D/o.\$)" = 8 ] || { echo "output sizes wrong"; bad=1; }
rm -rf $D
echo "cli_args: $( [ $bad = 0 ] && echo ok || echo FAILED )"
exit $bad
