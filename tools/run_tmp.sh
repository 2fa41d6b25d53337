python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
