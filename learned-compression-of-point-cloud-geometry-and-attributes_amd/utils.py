"""Small host-side helpers mirroring the reference's utils.py functions that sit on the path."""


def count_bits(strings):
    """Total bits of a nested list of byte strings (reference: utils.py:30-51; bpp = count_bits / N,
    train.py:268)."""
    total = 0
    for s in strings:
        total += count_bits(s) if isinstance(s, list) else len(s) * 8
    return total
