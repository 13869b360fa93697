#!/bin/bash
# round 5: full GPU suite on the tree with the product runners + regenerated train fixtures, smoke, the driver's bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
