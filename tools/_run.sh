export TMPDIR=/tmp
T2O_OWN_WGRAD=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short -x -k "actor or train or conv or golden" 2>&1 | tail -8
