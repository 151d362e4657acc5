"""Verification services for verifiable rewards: a batched async client to a remote sandbox
(``FUNCTIONCALL_SERVICE_DOMAIN``) with local fall-backs for math and code."""
