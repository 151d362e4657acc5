from . import mock_train, native  # noqa: F401
