"""Levenshtein distance over arbitrary sequences: what the reference gets from the un-vendored
``editdistance`` package (evaluation.py:41,43: ``editdistance.eval(expected, decoded)``)."""


def eval(a, b):  # noqa: A001  (name kept: callers write editdistance.eval(...))
  a, b = list(a), list(b)
  if len(a) < len(b):
    a, b = b, a
  row = list(range(len(b) + 1))
  for i, item_a in enumerate(a, start=1):
    diag, row[0] = row[0], i
    for j, item_b in enumerate(b, start=1):
      diag, row[j] = row[j], min(row[j] + 1, row[j - 1] + 1, diag + (item_a != item_b))
  return row[-1]
