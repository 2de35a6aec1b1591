function imageBlended = linearBlending(warpedImages, warpedWeights)
    %LINEARBLENDING Shadows PP/blending/linearBlending.m (integer inputs are rounded/saturated back to their class).
    if isempty(warpedImages), imageBlended = []; return; end
    cls = class(warpedImages{1});
    Ci = cellfun(@(x) single(gather(x)), warpedImages, 'UniformOutput', false);
    Wi = cellfun(@(x) single(gather(x(:, :, 1))), warpedWeights, 'UniformOutput', false);
    F = aps_mex('linear_blend', Ci, Wi);
    if size(warpedImages{1}, 3) == 1, F = F(:, :, 1); end
    if isinteger(warpedImages{1})
        imageBlended = cast(round(max(double(intmin(cls)), min(double(intmax(cls)), double(F)))), cls);
    else
        imageBlended = cast(F, cls);
    end
end
