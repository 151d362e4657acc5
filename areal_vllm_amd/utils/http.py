"""Async HTTP helpers with retry (``areal/utils/http.py`` in the reference)."""
import asyncio
from typing import Any, Dict, Optional

import aiohttp

from . import logging

logger = logging.getLogger("http")
DEFAULT_RETRIES = 3
DEFAULT_REQUEST_TIMEOUT = 3600


def get_default_connector():
    return aiohttp.TCPConnector(limit=0, use_dns_cache=False, force_close=True)


async def arequest_with_retry(addr: str, endpoint: str, payload: Optional[Dict[str, Any]] = None,
                              session: Optional[aiohttp.ClientSession] = None, method: str = "POST",
                              max_retries: Optional[int] = None, timeout: Optional[float] = None,
                              retry_delay: float = 1.0, verbose: bool = False) -> Any:
    timeout = timeout or DEFAULT_REQUEST_TIMEOUT
    max_retries = max_retries or DEFAULT_RETRIES
    url = f"http://{addr}{endpoint}"
    last_exc = None
    own = session is None
    if own:
        session = aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=timeout, sock_connect=timeout),
                                        read_bufsize=1024 * 1024 * 10, connector=get_default_connector())
    try:
        for attempt in range(max_retries):
            try:
                ctx_timeout = aiohttp.ClientTimeout(total=timeout)
                if method.upper() == "GET":
                    ctx = session.get(url, timeout=ctx_timeout)
                elif method.upper() == "POST":
                    ctx = session.post(url, json=payload, timeout=ctx_timeout)
                elif method.upper() == "PUT":
                    ctx = session.put(url, json=payload, timeout=ctx_timeout)
                elif method.upper() == "DELETE":
                    ctx = session.delete(url, timeout=ctx_timeout)
                else:
                    raise ValueError(f"Unsupported HTTP method: {method}")
                async with ctx as resp:
                    resp.raise_for_status()
                    ctype = resp.headers.get("Content-Type", "")
                    if "json" in ctype:
                        return await resp.json()
                    return await resp.text()
            except (aiohttp.ClientError, asyncio.TimeoutError) as e:
                last_exc = e
                if verbose:
                    logger.warning(f"request to {url} failed ({type(e).__name__}: {e}), attempt {attempt + 1}")
                if attempt < max_retries - 1:
                    await asyncio.sleep(retry_delay)
        raise RuntimeError(f"Failed after {max_retries} retries. Endpoint: {url}. Payload: "
                           f"{str(payload)[:200]}. Last error: {repr(last_exc)}")
    finally:
        if own:
            await session.close()


def response_ok(http_code: int) -> bool:
    return http_code == 200


def response_retryable(http_code: int) -> bool:
    return http_code in (408, 429, 500, 502, 503, 504)
