# the round-end checks: the whole -m gpu suite, then smoke()
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
