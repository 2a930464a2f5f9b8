"""Small helpers (envpool/python/utils.py)."""


def check_key_duplication(cls: str, keytype: str, keys: list) -> None:
    """Raise SystemError on duplicated spec keys."""
    ukeys, dup_keys = [], []
    for k in keys:
        if k in ukeys:
            dup_keys.append(k)
        else:
            ukeys.append(k)
    if len(dup_keys) > 0:
        raise SystemError(f"{cls} c++ code error. {keytype} keys {dup_keys} are duplicated.")
