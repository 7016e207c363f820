"""Small host-side helpers mirroring the reference's utils.py functions that sit on the path."""


def count_bits(strings):
    """Total bits of a nested list of byte strings (reference: utils.py:30-51; bpp = count_bits / N,
    train.py:268)."""
    total = 0
    for s in strings:
        total += count_bits(s) if isinstance(s, list) else len(s) * 8
    return total


def sparse_collate(coords, feats, device=None):
    """ME.utils.collation.sparse_collate (train.py:185-187): lists of per-sample coordinates [N_i, 3] and
    features [N_i, C] -> (int32 [sum N_i, 4] with the sample index in column 0, float32 [sum N_i, C])."""
    import torch
    cs, fs = [], []
    for b, (c, f) in enumerate(zip(coords, feats)):
        c = torch.as_tensor(c)
        f = torch.as_tensor(f)
        if c.shape[0] != f.shape[0]:
            raise ValueError("sample %d: %d coordinates for %d feature rows" % (b, c.shape[0], f.shape[0]))
        cs.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32), torch.floor(c.double()).to(torch.int32)], dim=1))
        fs.append(f.to(torch.float32))
    C, F = torch.cat(cs, dim=0), torch.cat(fs, dim=0)
    if device is not None:
        C, F = C.to(device), F.to(device)
    return C, F
