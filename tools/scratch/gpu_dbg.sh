cd $GRAFT_REPO_ROOT
for cfg in "SINGA_DIST_STRICT=0" "SINGA_DIST_STRICT=3" ; do
echo "== $cfg"
env $cfg DBG_CORR=1 DBG_THR=100 SINGA_DIST_CAPTURE=0 timeout -k 10 100 python -u tools/scratch/dbg_sparse.py 2>&1 | grep -a "segments\|eager\|Error\|acc" | head -5
done
echo "== threshold mode"
DBG_TOPK=0 DBG_CORR=1 DBG_THR=100 SINGA_DIST_CAPTURE=0 timeout -k 10 100 python -u tools/scratch/dbg_sparse.py 2>&1 | grep -a "segments\|eager\|Error\|acc" | head -5
true
