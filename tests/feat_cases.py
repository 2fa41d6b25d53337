"""Feature-extraction parity cases shared by the oracle tests (CPU) and the HIP tests (GPU).

HTK_CASES restate the option blocks of the reference's golden-vector tests
(src/feat/feature-fbank-test.cc:129-140,211-221,291-305,377-391 and
 src/feat/feature-mfcc-test.cc:130-140,214-224,298-309,383-392,466-479,554-565) with their
tolerances (:161,:242,:326,:412 -> 1e-3,1e-3,1e-3,1e-2; MFCC :164.. -> 1.0, we assert 5e-2) and
row range 10..N-10.  REF_CASES are command lines run through the reference binaries by
tests/golden/make_golden_feat.py."""
HTK_FBANK = {
    1: (dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_compat=1, htk_mode=1, use_energy=0), 1e-3),
    2: (dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=25.0, htk_compat=1, htk_mode=1, use_energy=0), 1e-3),
    3: (dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=25.0, htk_compat=1, htk_mode=1, use_energy=0, vtln_low=100.0, vtln_high=7500.0, vtln_warp=0.9), 1e-3),
    4: (dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=25.0, htk_compat=1, htk_mode=1, use_energy=0, vtln_low=100.0, vtln_high=7500.0, vtln_warp=1.1), 1e-2),
}
HTK_MFCC = {
    1: dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_mode=1, htk_compat=1, use_energy=0),
    2: dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_mode=1, htk_compat=1, use_energy=1),
    3: dict(dither=0.0, preemph_coeff=0.0, window_type="hamming", remove_dc_offset=0, htk_compat=1, use_energy=1, low_freq=20.0, htk_mode=1),
    4: dict(dither=0.0, window_type="hamming", remove_dc_offset=0, low_freq=0.0, htk_compat=1, use_energy=1, htk_mode=1),
    5: dict(dither=0.0, window_type="hamming", remove_dc_offset=0, htk_compat=1, use_energy=1, low_freq=0.0, vtln_low=100.0, vtln_high=7500.0, htk_mode=1, vtln_warp=1.1),
    6: dict(dither=0.0, preemph_coeff=0.97, window_type="hamming", remove_dc_offset=0, num_bins=24, low_freq=125.0, high_freq=7800.0, htk_compat=1, use_energy=0),
}
# name -> (kind, opts overrides, wav key)
REF_CASES = {
    "fbank_default40": ("fbank", dict(dither=0.0, num_bins=40), "wav"),
    "fbank_default23": ("fbank", dict(dither=0.0), "wav"),
    "fbank_energy_nosnip": ("fbank", dict(dither=0.0, num_bins=40, use_energy=1, snip_edges=0), "wav"),
    "fbank_hamming_nopow": ("fbank", dict(dither=0.0, window_type="hamming", use_power=0, remove_dc_offset=0, raw_energy=0, use_energy=1), "wav"),
    "mfcc_default": ("mfcc", dict(dither=0.0), "wav"),
    "mfcc_hires": ("mfcc", dict(dither=0.0, num_bins=40, num_ceps=40, low_freq=20.0, high_freq=-400.0, use_energy=0), "wav"),
    "mfcc_htkcompat": ("mfcc", dict(dither=0.0, htk_compat=1, use_energy=0, snip_edges=0), "wav"),
    "fbank_syn40": ("fbank", dict(dither=0.0, num_bins=40), "syn_wav"),
    "mfcc_syn_hires": ("mfcc", dict(dither=0.0, num_bins=40, num_ceps=40, low_freq=20.0, high_freq=-400.0, use_energy=0), "syn_wav"),
}
# windows that pad to 256 / 1024 samples (tests/golden/make_golden_feat_fftsizes.py): name -> (kind, opts, sample rate, samples, seed)
FFTSIZE_CASES = {
    "fbank_8k": ("fbank", dict(dither=0.0, samp_freq=8000.0, num_bins=23), 8000, 11000, 41),
    "mfcc_8k": ("mfcc", dict(dither=0.0, samp_freq=8000.0), 8000, 9001, 42),
    "mfcc_8k_nosnip": ("mfcc", dict(dither=0.0, samp_freq=8000.0, snip_edges=0, use_energy=0, low_freq=40.0, high_freq=-200.0), 8000, 7333, 43),
    "fbank_32k": ("fbank", dict(dither=0.0, samp_freq=32000.0, num_bins=40), 32000, 40000, 44),
    "fbank_16k_50ms": ("fbank", dict(dither=0.0, frame_length_ms=50.0, num_bins=40, use_energy=1), 16000, 20000, 45),
    "mfcc_16k_40ms_hires": ("mfcc", dict(dither=0.0, frame_length_ms=40.0, frame_shift_ms=15.0, num_bins=40, num_ceps=40, low_freq=20.0, high_freq=-400.0, use_energy=0), 16000, 24000, 46),
}

# files whose rate differs from --sample-frequency (OfflineFeatureTpl::ComputeFeatures' resampling branch, feat/feature-common-inl.h:29-57): (kind, options, file rate, samples, seed)
RESAMPLE_CASES = {
    "down_16k_to_8k": ("fbank", dict(dither=0.0, samp_freq=8000.0, num_bins=23), 16000, 20011, 51),
    "up_8k_to_16k": ("fbank", dict(dither=0.0, samp_freq=16000.0, num_bins=40), 8000, 9001, 52),
    "down_44k1_to_16k": ("mfcc", dict(dither=0.0, samp_freq=16000.0), 44100, 30000, 53),
    "down_22k05_to_16k_nosnip": ("fbank", dict(dither=0.0, samp_freq=16000.0, num_bins=40, snip_edges=0), 22050, 15001, 54),
}
