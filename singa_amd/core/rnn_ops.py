"""Recurrent network handle and ops (filled in by the RNN milestone)."""
__all__ = []
