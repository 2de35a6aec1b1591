function F = multiBandBlending(Ci, Wi, levels, onGPU, sigma) %#ok<INUSL>
    %MULTIBANDBLENDING Shadows PP/blending/multiBandBlending.m (same signature; onGPU is ignored: always device).
    if nargin < 5, sigma = 1.0; end
    gray = size(Ci{1}, 3) == 1;
    for k = 1:numel(Ci)
        Ci{k} = single(gather(Ci{k})); Wi{k} = single(gather(Wi{k}));
    end
    F = aps_mex('multiband_blend', Ci, Wi, double(levels), double(sigma));
    if gray, F = F(:, :, 1); end
end
