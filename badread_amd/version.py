"""Version string printed by the CLI banner (mirrors /root/reference/badread/version.py:17)."""
__version__ = '0.4.2'
__backend__ = 'MI355X-native (HIP gfx950)'
