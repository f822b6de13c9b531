"""Stand-alone codec entry points (reference: src/ai_pcc/GausPcgc/compress_ue_4stage_conv.py,
decompress_ue_4stage_conv.py):  python -m gauspcc_amd.cli.compress / .decompress"""
