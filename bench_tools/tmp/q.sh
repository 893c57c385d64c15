cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q 2>&1 | tail -6
