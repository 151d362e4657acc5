"""HTTP client helpers for the control plane (trainer -> rollout servers, workers -> managers).

``arequest_with_retry`` issues one JSON request and retries transport failures and time-outs; HTTP error
statuses count as failures too. It opens a session of its own when the caller has none to share. Role of
``areal/utils/http.py`` in the reference.
"""
import asyncio
import contextlib
from typing import Any, Dict, Optional

import aiohttp

from . import logging

logger = logging.getLogger("http")
DEFAULT_RETRIES = 3
DEFAULT_REQUEST_TIMEOUT = 3600
_VERBS_WITH_BODY = {"POST", "PUT"}
_VERBS = _VERBS_WITH_BODY | {"GET", "DELETE"}
_RETRYABLE_STATUS = frozenset({408, 429, 500, 502, 503, 504})


def get_default_connector() -> aiohttp.TCPConnector:
    # unlimited connections, no keep-alive: servers restart under us during recover runs
    return aiohttp.TCPConnector(limit=0, use_dns_cache=False, force_close=True)


@contextlib.asynccontextmanager
async def _session_for(shared: Optional[aiohttp.ClientSession], timeout: float):
    if shared is not None:
        yield shared
        return
    async with aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=timeout, sock_connect=timeout),
                                     read_bufsize=10 << 20, connector=get_default_connector()) as own:
        yield own


async def _body(resp: aiohttp.ClientResponse) -> Any:
    resp.raise_for_status()
    if "json" in resp.headers.get("Content-Type", ""):
        return await resp.json()
    return await resp.text()


async def arequest_with_retry(addr: str, endpoint: str, payload: Optional[Dict[str, Any]] = None,
                              session: Optional[aiohttp.ClientSession] = None, method: str = "POST",
                              max_retries: Optional[int] = None, timeout: Optional[float] = None,
                              retry_delay: float = 1.0, verbose: bool = False) -> Any:
    verb = method.upper()
    if verb not in _VERBS:
        raise ValueError(f"Unsupported HTTP method: {method}")
    budget = timeout or DEFAULT_REQUEST_TIMEOUT
    attempts = max_retries or DEFAULT_RETRIES
    url = f"http://{addr}{endpoint}"
    kwargs: Dict[str, Any] = dict(timeout=aiohttp.ClientTimeout(total=budget))
    if verb in _VERBS_WITH_BODY:
        kwargs["json"] = payload
    failure: Optional[BaseException] = None
    async with _session_for(session, budget) as s:
        for n in range(1, attempts + 1):
            try:
                async with s.request(verb, url, **kwargs) as resp:
                    return await _body(resp)
            except (aiohttp.ClientError, asyncio.TimeoutError) as e:
                failure = e
                if verbose:
                    logger.warning(f"{verb} {url}: {type(e).__name__}: {e} (attempt {n} of {attempts})")
            if n < attempts:
                await asyncio.sleep(retry_delay)
    raise RuntimeError(f"Failed after {attempts} retries. Endpoint: {url}. Payload: {str(payload)[:200]}. "
                       f"Last error: {failure!r}")


def response_ok(http_code: int) -> bool:
    return http_code == 200


def response_retryable(http_code: int) -> bool:
    return http_code in _RETRYABLE_STATUS
