export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -4
bash tools/ab_train.sh "T2O_OWN_CONV=wfds" "T2O_OWN_WGRAD=0" "T2O_OWN_CONV=wfds"
