cd $GRAFT_REPO_ROOT
timeout 500 python -X faulthandler -m pytest tests -m gpu -q --timeout 200 2>&1 | grep -v "^  File\|^Extension" | tail -25
