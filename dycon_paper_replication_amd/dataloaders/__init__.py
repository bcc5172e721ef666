"""Data path mirror."""
