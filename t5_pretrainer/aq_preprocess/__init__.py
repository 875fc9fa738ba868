"""alias package, see t5_pretrainer/__init__.py"""
