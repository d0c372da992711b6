"""Mirror of the reference's utils/__init__.py (utils/__init__.py:4-19)."""
import torch


def underline(content, style='-'):
    """Underlining a sentence."""
    return content + '\n' + style * len(content.strip())


def empty_tensor():
    """The reference's "empty" sentinel: a 0-dim tensor(0)."""
    return torch.tensor(0)


def is_empty_tensor(t):
    return len(t.size()) == 0
