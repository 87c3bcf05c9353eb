"""Exception type raised by the semantic store — counterpart of the reference's
`ValidationError` (codd_engine/validation_engine/metrics/validation_result.py:86-89)."""


class ValidationError(Exception):
    """Input rejected by the store's validation rules."""
