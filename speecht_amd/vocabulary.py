"""Character vocabulary of speechT (mirror of speecht/vocabulary.py:16-81).

a-z -> 0..25, apostrophe -> 26, space -> 27; SIZE = 28.  The CTC blank is SIZE (num_classes =
SIZE + 1, speech_model.py:301).  Table-driven restatement; pinned by tests/golden/
vocabulary_golden.json, which was produced by importing the reference module itself.
"""
import string

_LETTERS = string.ascii_lowercase + "' "
_TO_ID = {ch: i for i, ch in enumerate(_LETTERS)}

APOSTROPHE = _TO_ID["'"]
SPACE_ID = _TO_ID[' ']
SIZE = len(_LETTERS)
A_ASCII_CODE = ord('a')


def letter_to_id(letter):
  try:
    return _TO_ID[letter]
  except KeyError:
    # the reference computes ord(letter) - ord('a') for anything else (vocabulary.py:39)
    return ord(letter) - A_ASCII_CODE


def id_to_letter(identifier):
  identifier = int(identifier)
  if 0 <= identifier < SIZE:
    return _LETTERS[identifier]
  return chr(identifier + A_ASCII_CODE)


def sentence_to_ids(sentence):
  return [letter_to_id(letter) for letter in sentence.lower()]


def ids_to_sentence(identifiers):
  return ''.join(id_to_letter(i) for i in identifiers)
